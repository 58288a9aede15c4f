"""Engine: Python mirror of the reference's interface for the hot path.

One method per C-ABI entry point of include/ltrans_hip.h; method names follow
the reference procedures they stand in for (``update_particles`` =
LTRANS.f90:707).  The default library is the in-tree HIP build
``csrc/libltrans_hip.so``; it is REQUIRED -- there is no CPU fallback and the
load fails loudly when the library is missing.  ``Engine(cfg)`` always binds that
library; the test suite derives its own class for the CPU oracle (tests/oracle_lib.py).
"""
import ctypes as C
import os

import numpy as np

from . import abi
from .abi import (LtBounds, LtConfig, LtError, LtGrid, LtHabitat, LtParticles, LtSlab,
                  LtStepStats, PARTICLE_FIELDS)

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIBRARY = os.path.join(_HERE, "csrc", "libltrans_hip.so")
_hip_lib = None


def load_hip_library(path=None):
    """dlopen the HIP engine.  Raises if it has not been built (no fallback)."""
    global _hip_lib
    if path is None and _hip_lib is not None:
        return _hip_lib
    p = path or os.environ.get("LT_HIP_LIBRARY") or HIP_LIBRARY
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'); "
            "this engine has no CPU fallback")
    lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
    if path is None:
        _hip_lib = lib
    return lib


def _f8(a, n=None, name="array"):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel(order="K"))
    if n is not None and a.size != n:
        raise ValueError(f"{name}: expected {n} values, got {a.size}")
    return a


def _i4(a, n=None, name="array"):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.int32).ravel(order="K"))
    if n is not None and a.size != n:
        raise ValueError(f"{name}: expected {n} values, got {a.size}")
    return a


def _fcol(a, dtype):
    """Fortran (column-major) flattening of a 1-D/2-D array."""
    a = np.asarray(a, dtype=dtype)
    return np.ascontiguousarray(a.ravel(order="F"))


def _dp(a):
    return a.ctypes.data_as(abi.c_dp) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(abi.c_ip) if a is not None else None


def create_bounds(vi, uj, mask_rho, x_u, y_u, x_v, y_v, lib=None):
    """createBounds (boundary_module.f90:82-1215) on the host side of the library: -> the dict lt_set_bounds takes."""
    lib = lib or load_hip_library()
    m = np.asfortranarray(np.asarray(mask_rho, dtype=np.int32))
    arr = [np.asfortranarray(np.asarray(a, dtype=np.float64)) for a in (x_u, y_u, x_v, y_v)]
    cap = int(vi) * int(uj)
    n = [C.c_int32() for _ in range(4)]
    bx, by, hx, hy = (np.zeros(cap) for _ in range(4))
    hid, land = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.int32)
    bnd_x, bnd_y = np.zeros(2 * cap), np.zeros(2 * cap)
    f = lib.lt_create_bounds
    f.restype = C.c_int
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = f(C.c_int32(int(vi)), C.c_int32(int(uj)), vp(m), *[vp(a) for a in arr], C.c_int32(cap), *[C.byref(v) for v in n],
           vp(bx), vp(by), vp(hx), vp(hy), vp(hid), vp(bnd_x), vp(bnd_y), vp(land))
    if rc != 0:
        raise LtError(rc, "lt_create_bounds")
    nb, mb, mi, ni = (v.value for v in n)
    return dict(nbounds=nb, maxbound=mb, maxisland=mi, numislands=ni,
                bnd_x=np.asfortranarray(bnd_x[:2 * nb].reshape(2, nb, order="F")),
                bnd_y=np.asfortranarray(bnd_y[:2 * nb].reshape(2, nb, order="F")), land=land[:nb].copy(),
                bx=bx[:mb].copy(), by=by[:mb].copy(), hx=hx[:mi].copy(), hy=hy[:mi].copy(), hid=hid[:mi].copy())


class Engine:
    """Device-resident replacement of the module state + update_particles()."""

    def __init__(self, cfg: LtConfig):
        self._open(cfg, load_hip_library(), "lt_")

    def _open(self, cfg, lib, prefix):
        """bind the entry points of one implementation of the ABI and create the engine"""
        self.lib = lib
        self.prefix = prefix
        self.cfg = cfg.copy()
        self._h = C.c_void_p()
        self._fn = {}
        for name in abi.ENTRY_POINTS:
            f = getattr(self.lib, prefix + name)
            f.restype = C.c_int
            self._fn[name] = f
        rc = self._fn["create"](C.byref(self.cfg), C.byref(self._h))
        if rc != 0:
            self._h = C.c_void_p()
            raise LtError(rc, prefix + "create")
        self.numpar = int(cfg.numpar)
        self.us, self.ws = int(cfg.us), int(cfg.ws)
        self._grid = None

    # -- lifecycle ---------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._fn["destroy"](self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, where):
        if rc < 0:
            detail = ""
            if self.prefix == "lt_":
                f = self.lib.lt_last_error
                f.restype = C.c_char_p
                msg = f(self._h)
                detail = msg.decode() if msg else ""
            raise LtError(rc, self.prefix + where, detail)
        return rc

    # -- uploads -----------------------------------------------------------
    def set_grid(self, g):
        """g: mapping with the arrays initGrid builds (hydrodynamic_module.f90:88-657);
        RE/UE/VE are (4,elements), *_Adjacent (elements,10) in Fortran index order."""
        s = LtGrid()
        keep = []
        for k in ("rho_nodes", "u_nodes", "v_nodes", "rho_elements", "u_elements", "v_elements"):
            setattr(s, k, int(g[k]))
        sizes = {"rx": s.rho_nodes, "ry": s.rho_nodes, "ux": s.u_nodes, "uy": s.u_nodes,
                 "vx": s.v_nodes, "vy": s.v_nodes, "depth": s.rho_nodes,
                 "rho_angle": s.rho_nodes, "SC": self.us, "CS": self.us,
                 "SCW": self.ws, "CSW": self.ws}
        for k, n in sizes.items():
            a = _f8(g[k], n, k)
            keep.append(a)
            setattr(s, k, _dp(a))
        for k, n in (("rho_mask", s.rho_nodes), ("u_mask", s.u_nodes), ("v_mask", s.v_nodes)):
            if g.get(k) is not None:
                a = _i4(g[k], n, k)
                keep.append(a)
                setattr(s, k, _ip(a))
        for k, n in (("RE", s.rho_elements), ("UE", s.u_elements), ("VE", s.v_elements)):
            a = _fcol(np.asarray(g[k]).reshape(4, n, order="F"), np.int32)
            keep.append(a)
            setattr(s, k, _ip(a))
        for k, n in (("r_Adjacent", s.rho_elements), ("u_Adjacent", s.u_elements),
                     ("v_Adjacent", s.v_elements)):
            a = _fcol(np.asarray(g[k]).reshape(n, 10, order="F"), np.int32)
            keep.append(a)
            setattr(s, k, _ip(a))
        self._check(self._fn["set_grid"](self._h, C.byref(s)), "set_grid")
        self._grid = {k: int(g[k]) for k in ("rho_nodes", "u_nodes", "v_nodes", "rho_elements")}

    def set_bounds(self, b):
        s = LtBounds()
        keep = []
        s.nbounds, s.maxbound = int(b["nbounds"]), int(b["maxbound"])
        s.maxisland, s.numislands = int(b["maxisland"]), int(b["numislands"])
        for k, n in (("bnd_x", 2 * s.nbounds), ("bnd_y", 2 * s.nbounds), ("bx", s.maxbound),
                     ("by", s.maxbound), ("hx", s.maxisland), ("hy", s.maxisland)):
            a = _fcol(b[k], np.float64)
            if a.size != n:
                raise ValueError(f"{k}: expected {n} values, got {a.size}")
            keep.append(a)
            setattr(s, k, _dp(a))
        for k, n in (("land", s.nbounds), ("hid", s.maxisland)):
            a = _i4(b[k], n, k)
            keep.append(a)
            setattr(s, k, _ip(a))
        self._check(self._fn["set_bounds"](self._h, C.byref(s)), "set_bounds")

    def set_habitat(self, h):
        s = LtHabitat()
        keep = []
        for k in ("pedges", "hedges", "minpolyid", "maxpolyid", "minholeid", "maxholeid"):
            setattr(s, k, int(h[k]))
        for k in ("polys", "holes", "maxbdis", "maxhdis"):
            a = _fcol(h[k], np.float64)
            keep.append(a)
            setattr(s, k, _dp(a))
        for k in ("polyspecs", "holespecs", "elepolys_ptr", "elepolys_ids", "polyholes_ptr",
                  "polyholes_ids"):
            a = _fcol(h[k], np.int32)
            keep.append(a)
            setattr(s, k, _ip(a))
        self._check(self._fn["set_habitat"](self._h, C.byref(s)), "set_habitat")

    # -- init-time geometry --------------------------------
    def build_adjacency(self, E, row):
        """initGrid's adjacency lists (hydrodynamic_module.f90:588-645) of the (4,elements)
        node table E; row = vi (rho, v) or ui (u).  Returns (elements,10), Fortran order."""
        E = _fcol(np.asarray(E), np.int32)
        n = E.size // 4
        adj = np.zeros((n, 10), dtype=np.int32, order="F")
        self._check(self._fn["build_adjacency"](C.c_int32(int(self.cfg.device)), C.c_int32(n), C.c_int32(int(row)),
                                                _ip(E), _ip(adj)), "build_adjacency")
        return adj

    def locate_particles(self, x, y):
        """setEle_all (hydrodynamic_module.f90:1536-1676).  Returns (r, u, v, err, par)."""
        x = _f8(x, name="x")
        y = _f8(y, x.size, "y")
        n = x.size
        r, u, v = (np.zeros(n, dtype=np.int32) for _ in range(3))
        err, par = C.c_int32(0), C.c_int32(0)
        self._check(self._fn["locate_particles"](self._h, C.c_int32(n), _dp(x), _dp(y), _ip(r), _ip(u), _ip(v),
                                                 C.byref(err), C.byref(par)), "locate_particles")
        return r, u, v, err.value, par.value

    def create_polyspecs(self, h):
        """createPolySpecs (settlement_module.f90:245-480) from the edge-point tables of h;
        installs the habitat.  Returns the lists as lt_set_habitat would take them."""
        s = LtHabitat()
        keep = []
        for k in ("pedges", "hedges", "minpolyid", "maxpolyid", "minholeid", "maxholeid"):
            setattr(s, k, int(h[k]))
        for k in ("polys", "holes", "maxbdis", "maxhdis"):
            a = _fcol(h[k], np.float64)
            keep.append(a)
            setattr(s, k, _dp(a))
        self._check(self._fn["create_polyspecs"](self._h, C.byref(s)), "create_polyspecs")
        npoly = s.maxpolyid - s.minpolyid + 1
        nhole = (s.maxholeid - s.minholeid + 1) if s.hedges > 0 else 0
        ne = self._grid["rho_elements"]
        out = dict(polyspecs=np.zeros((npoly, 2), dtype=np.int32, order="F"),
                   holespecs=np.zeros((max(nhole, 1), 2), dtype=np.int32, order="F"),
                   elepolys_ptr=np.zeros(ne + 1, dtype=np.int32),
                   polyholes_ptr=np.zeros(npoly + 1, dtype=np.int32))
        get = self._fn["get_polyspecs"]
        self._check(get(self._h, _ip(out["polyspecs"]), _ip(out["holespecs"]) if nhole else None,
                        _ip(out["elepolys_ptr"]), None, C.c_int32(0), _ip(out["polyholes_ptr"]), None,
                        C.c_int32(0)), "get_polyspecs")
        nl, nph = int(out["elepolys_ptr"][-1]), int(out["polyholes_ptr"][-1])
        out["elepolys_ids"] = np.zeros(max(nl, 1), dtype=np.int32)
        out["polyholes_ids"] = np.zeros(max(nph, 1), dtype=np.int32)
        self._check(get(self._h, None, None, None, _ip(out["elepolys_ids"]), C.c_int32(nl), None,
                        _ip(out["polyholes_ids"]), C.c_int32(nph)), "get_polyspecs")
        return out

    def set_particles(self, **arrays):
        """par=(numpar,13), P_r_element=..., ... (names of abi.PARTICLE_FIELDS)."""
        s = LtParticles()
        keep = []
        for name, _, dt, cols in PARTICLE_FIELDS:
            if arrays.get(name) is None:
                continue
            a = np.asarray(arrays[name])
            if cols > 1:
                a = a.reshape(self.numpar, cols)
            a = _fcol(a, np.dtype(dt))
            if a.size != self.numpar * cols:
                raise ValueError(f"{name}: expected {self.numpar * cols} values, got {a.size}")
            keep.append(a)
            setattr(s, name, _dp(a) if dt == "f8" else _ip(a))
        unknown = set(arrays) - {f[0] for f in PARTICLE_FIELDS}
        if unknown:
            raise KeyError(f"unknown particle arrays {sorted(unknown)}")
        self._check(self._fn["set_particles"](self._h, C.byref(s)), "set_particles")

    def get_output(self, reset_hits=False, salt_temp=False):
        """lt_get_output: dict lon, lat, z, age, status (+ salt, temp), hitBottom, hitLand."""
        n = self.numpar
        o = abi.LtOutput()
        res = {}
        for k in ("lon", "lat", "z", "age", "status") + (("salt", "temp") if salt_temp else ()):
            res[k] = np.zeros(n, dtype=np.float64)
            setattr(o, k, _dp(res[k]))
        for k in ("hitBottom", "hitLand"):
            res[k] = np.zeros(n, dtype=np.int32)
            setattr(o, k, _ip(res[k]))
        self._check(self._fn["get_output"](self._h, C.byref(o), int(bool(reset_hits))), "get_output")
        return res

    def set_slab(self, slot, zeta, Uvel, Vvel, Wvel, Kh=None, salt=None, temp=None):
        """One masked time record into slot 1..3 (updateHydro, hydrodynamic_module.f90:1371)."""
        g = self._grid
        if g is None:
            raise LtError(-3, "set_slab", "set_grid first")
        s = LtSlab()
        z = _f8(zeta, g["rho_nodes"], "zeta")
        u = _fcol(Uvel, np.float64)
        v = _fcol(Vvel, np.float64)
        w = _fcol(Wvel, np.float64)
        if u.size != g["u_nodes"] * self.us or v.size != g["v_nodes"] * self.us \
                or w.size != g["rho_nodes"] * self.ws:
            raise ValueError("slab extents do not match the grid")
        s.zeta, s.Uvel, s.Vvel, s.Wvel = _dp(z), _dp(u), _dp(v), _dp(w)
        k = None
        if Kh is not None:
            k = _fcol(Kh, np.float64)
            if k.size != g["rho_nodes"] * self.ws:
                raise ValueError("Kh extent does not match the grid")
            s.Kh = _dp(k)
        st = []
        for name, a in (("salt", salt), ("temp", temp)):
            if a is not None:
                a = _fcol(a, np.float64)
                if a.size != g["rho_nodes"] * self.us:
                    raise ValueError(f"{name} extent does not match the grid")
                st.append(a)
                setattr(s, name, _dp(a))
        self._check(self._fn["set_slab"](self._h, int(slot), C.byref(s)), "set_slab")

    def slab_prefetch(self, rom, dims, masked=False, window=None):
        """lt_slab_prefetch: ``rom`` = dict romZ (vi,uj), romU (ui,uj,us), romV (vi,vj,us),
        romW / romKH (vi,uj,ws) as the host read them (float64 or float32, all the same
        type, Fortran order); ``dims`` = (vi, uj, ui, vj); ``window`` = t_ijruv(12)."""
        vi, uj, ui, vj = (int(d) for d in dims)
        dt = np.dtype(rom["romZ"].dtype)
        if dt not in (np.dtype(np.float64), np.dtype(np.float32)):
            raise ValueError("rom arrays must be float64 or float32")
        shapes = {"romZ": (vi, uj), "romU": (ui, uj, self.us), "romV": (vi, vj, self.us),
                  "romW": (vi, uj, self.ws), "romKH": (vi, uj, self.ws),
                  "romS": (vi, uj, self.us), "romT": (vi, uj, self.us)}
        r = abi.LtRomRecord()
        r.struct_size = C.sizeof(abi.LtRomRecord)
        r.dtype = abi.LT_ROM_F32 if dt == np.dtype(np.float32) else abi.LT_ROM_F64
        r.vi, r.uj, r.ui, r.vj = vi, uj, ui, vj
        r.masked = 1 if masked else 0
        r.has_window = 0 if window is None else 1
        if window is not None:
            for i, v in enumerate(window):
                r.window[i] = int(v)
        keep = []
        for name, shp in shapes.items():
            a = rom.get(name)
            if a is None:
                continue
            a = np.asfortranarray(a, dtype=dt)
            if a.shape != shp:
                raise ValueError(f"{name} has shape {a.shape}, expected {shp}")
            keep.append(a)
            setattr(r, name, a.ctypes.data)
        self._check(self._fn["slab_prefetch"](self._h, C.byref(r)), "slab_prefetch")

    def slab_commit(self, t_b, t_c, t_f):
        self._check(self._fn["slab_commit"](self._h, int(t_b), int(t_c), int(t_f)), "slab_commit")

    def set_time_slots(self, t_b, t_c, t_f):
        self._check(self._fn["set_time_slots"](self._h, int(t_b), int(t_c), int(t_f)),
                    "set_time_slots")

    def set_rng(self, mt, mti):
        a = np.ascontiguousarray(np.asarray(mt, dtype=np.uint32)).view(np.int32)
        self._check(self._fn["set_rng"](self._h, _ip(a), int(mti)), "set_rng")

    def get_rng(self):
        a = np.zeros(624, dtype=np.int32)
        mti = C.c_int32()
        self._check(self._fn["get_rng"](self._h, _ip(a), C.byref(mti)), "get_rng")
        return a.view(np.uint32), mti.value

    def count_active(self, ix):
        ix = _f8(ix, 3, "ix")
        n = C.c_int64()
        self._check(self._fn["count_active"](self._h, _dp(ix), C.byref(n)), "count_active")
        return n.value

    def count_draws(self, ex, ix, p, it):
        """lt_count_draws: runs the first part of the step; the next update_particles with
        the same arguments continues from there."""
        ex = np.ascontiguousarray(ex, dtype=np.float64)
        ix = np.ascontiguousarray(ix, dtype=np.float64)
        n = C.c_int64(0)
        self._check(self._fn["count_draws"](self._h, _dp(ex), _dp(ix), int(p), int(it), C.byref(n)),
                    "count_draws")
        return int(n.value)

    def set_rng_window_draws(self, draws_before, draws_total):
        self._check(self._fn["set_rng_window_draws"](self._h, C.c_int64(int(draws_before)),
                                                     C.c_int64(int(draws_total))), "set_rng_window_draws")

    def exchange_rng_window(self, ex, ix, p, it):
        """lt_exchange_rng_window: the first part of the coming step, the all-gather of the shards' draw counts on
        the engine's communicator (comm_init_file) and this shard's window; update_particles with the same
        arguments continues from there."""
        ex = _f8(ex, 3, "ex")
        ix = _f8(ix, 3, "ix")
        f = self.lib.lt_exchange_rng_window
        f.restype = C.c_int
        self._check(f(self._h, _dp(ex), _dp(ix), int(p), int(it)), "exchange_rng_window")

    def source_hash(self):
        f = self.lib.lt_source_hash
        f.restype = C.c_char_p
        return f().decode()

    def set_rng_window(self, active_before, active_total, persistent=False):
        if persistent:
            f = self.lib.lt_set_rng_window_static
            f.restype = C.c_int
            self._check(f(self._h, C.c_int64(int(active_before)), C.c_int64(int(active_total))), "set_rng_window_static")
            return
        return self._set_rng_window(active_before, active_total)

    def _set_rng_window(self, active_before, active_total):
        self._check(self._fn["set_rng_window"](self._h, C.c_int64(int(active_before)),
                                               C.c_int64(int(active_total))), "set_rng_window")

    # -- the hot path --------------------------------------------------------
    def update_particles(self, ex, ix, p, it):
        """= update_particles() of LTRANS.f90:707 for main's ex, ix, p, it."""
        ex = _f8(ex, 3, "ex")
        ix = _f8(ix, 3, "ix")
        return self._check(self._fn["step"](self._h, _dp(ex), _dp(ix), int(p), int(it)), "step")

    step = update_particles

    def sync(self):
        self._check(self._fn["sync"](self._h), "sync")

    def stats(self):
        s = LtStepStats()
        self._check(self._fn["get_stats"](self._h, C.byref(s)), "get_stats")
        return {"active": s.active, "errors": s.errors, "rng_draws": s.rng_draws}

    # -- downloads -----------------------------------------------------------
    def get_particles(self, fields=None, out=None):
        """-> {name: array}.  `out`: the dictionary an earlier call returned, to download into the same arrays (a host
        that prints every few steps keeps its arrays; 240 MB of fresh pages per call cost as much as the download)"""
        s = LtParticles()
        reuse, out = out, {}
        for name, _, dt, cols in PARTICLE_FIELDS:
            if fields is not None and name not in fields:
                continue
            a = None
            if reuse is not None and name in reuse:
                a = reuse[name].ravel(order="F")
                if a.base is None or a.size != self.numpar * cols or a.dtype != np.dtype(dt) or not a.flags.writeable:
                    a = None
            if a is None:
                a = np.zeros(self.numpar * cols, dtype=np.dtype(dt))
            out[name] = a
            setattr(s, name, _dp(a) if dt == "f8" else _ip(a))
        self._check(self._fn["get_particles"](self._h, C.byref(s)), "get_particles")
        for name, _, dt, cols in PARTICLE_FIELDS:
            if name in out and cols > 1:
                out[name] = out[name].reshape(self.numpar, cols, order="F")
        return out

    def pack_positions(self, dst_device_ptr):
        f = self.lib.lt_pack_positions
        f.restype = C.c_int
        self._check(f(self._h, C.c_void_p(int(dst_device_ptr))), "pack_positions")

    # -- multi-GPU: the RCCL all-gather behind the ABI ---------------------------
    def comm_init_file(self, path, nranks, rank, timeout_s=60):
        f = self.lib.lt_comm_init_file
        f.restype = C.c_int
        self._check(f(self._h, str(path).encode(), int(nranks), int(rank), int(timeout_s)), "comm_init_file")

    def comm_unique_id(self):
        """lt_comm_unique_id: RCCL's 128-byte id (rank 0 makes it, every rank passes it to comm_init)"""
        buf = (C.c_ubyte * 128)()
        f = self.lib.lt_comm_unique_id
        f.restype = C.c_int
        rc = f(buf)
        if rc != 0:
            raise LtError(rc, "lt_comm_unique_id")
        return bytes(buf)

    def comm_init(self, id128, nranks, rank):
        f = self.lib.lt_comm_init
        f.restype = C.c_int
        buf = (C.c_ubyte * 128).from_buffer_copy(bytes(id128))
        self._check(f(self._h, buf, int(nranks), int(rank)), "comm_init")

    def allgather_positions(self, counts):
        """-> xyz(sum(counts), 3) of all shards in rank order (host array)"""
        counts = _i4(counts)
        out = np.zeros((int(counts.sum()), 3), order="F")
        f = self.lib.lt_allgather_positions
        f.restype = C.c_int
        self._check(f(self._h, _ip(counts), _dp(out), None), "allgather_positions")
        return out

    def allgather_positions_device(self, counts, dst_device_ptr):
        """the same into device memory, asynchronously (allgather_wait() says when it is complete)"""
        counts = _i4(counts)
        f = self.lib.lt_allgather_positions
        f.restype = C.c_int
        self._check(f(self._h, _ip(counts), None, C.c_void_p(int(dst_device_ptr))), "allgather_positions")

    def allgather_wait(self):
        f = self.lib.lt_allgather_wait
        f.restype = C.c_int
        self._check(f(self._h), "allgather_wait")

    def comm_destroy(self):
        f = self.lib.lt_comm_destroy
        f.restype = C.c_int
        self._check(f(self._h), "comm_destroy")

    def profile_begin(self, max_steps):
        f = self.lib.lt_profile_begin
        f.restype = C.c_int
        self._check(f(self._h, int(max_steps)), "profile_begin")

    def profile_end(self):
        """-> (total kernel milliseconds, launches) measured with HIP events."""
        f = self.lib.lt_profile_end
        f.restype = C.c_int
        ms, n = C.c_double(), C.c_int32()
        self._check(f(self._h, C.byref(ms), C.byref(n)), "profile_end")
        return ms.value, n.value

    def counters(self):
        f = self.lib.lt_get_counters
        f.restype = C.c_int
        a = np.zeros(8, dtype=np.uint64)
        self._check(f(self._h, a.ctypes.data_as(C.c_void_p)), "get_counters")
        out = {"active": int(a[0]), "errors": int(a[1]), "sigerr": int(a[2]),
               "brute_bounds": int(a[3]), "pip_full": int(a[4]), "vt_queued": int(a[5]), "vt_rounds": int(a[6]),
               "rk_slow": int(a[7]), "resorts": 0, "last_resort_ms": 0.0}
        g = getattr(self.lib, "lt_get_resort_stats", None)     # (not in the oracle)
        if g is not None:
            g.restype = C.c_int
            n, ms = C.c_int64(), C.c_double()
            self._check(g(self._h, C.byref(n), C.byref(ms)), "get_resort_stats")
            out["resorts"], out["last_resort_ms"] = n.value, max(ms.value, 0.0)
        return out

    def stream(self):
        f = self.lib.lt_stream
        f.restype = C.c_void_p
        return f(self._h)
