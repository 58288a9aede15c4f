"""ctypes access to the CPU checkers under oracle/ (test infrastructure only)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libltrans_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libltrans_ref.so")

_vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731


def build_oracle():
    # (tests/test_sanitizers.py: the sanitizer build of the oracle, made by `make -C oracle asan`)
    if os.environ.get("LT_ORACLE_LIBRARY"):
        return os.environ["LT_ORACLE_LIBRARY"]
    srcs = [os.path.join(ORACLE_DIR, f) for f in
            ("ltrans_oracle.c", "ltrans_oracle.h", "ltrans_oracle_engine.inc", "ltrans_oracle_init.inc")]
    if (not os.path.exists(ORACLE_SO)
            or os.path.getmtime(ORACLE_SO) < max(os.path.getmtime(s) for s in srcs)):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "-B", "liboracle"])
    return ORACLE_SO


def load_oracle():
    lib = C.CDLL(build_oracle())
    for f in ("lto_genrand_real1", "lto_genrand_real3", "lto_norm", "lto_polintd", "lto_hval",
              "lto_hpval", "lto_getSlevel", "lto_getWlevel"):
        getattr(lib, f).restype = C.c_double
    lib.lto_genrand_int32.restype = C.c_uint32
    return lib


def load_ref():
    if not os.path.exists(REF_SO):
        return None
    lib = C.CDLL(REF_SO)
    for f in ("ref_polintd", "ref_hval", "ref_hpval"):
        getattr(lib, f).restype = C.c_double
    return lib


class MT(C.Structure):
    _fields_ = [("mt", C.c_uint32 * 624), ("mti", C.c_int32)]


# ---- uniform wrappers: same python signature for oracle ("o") and reference ("r")
class Leaf:
    def __init__(self, lib, kind):
        self.lib, self.kind = lib, kind
        self._mt = MT()

    def init_genrand(self, seed):
        if self.kind == "o":
            self.lib.lto_init_genrand(C.byref(self._mt), C.c_uint32(seed))
        else:
            self.lib.ref_init_genrand(C.c_int(np.int32(np.uint32(seed))))

    def int32(self, n):
        out = np.zeros(n, dtype=np.uint32)
        if self.kind == "o":
            for i in range(n):
                out[i] = self.lib.lto_genrand_int32(C.byref(self._mt))
        else:
            self.lib.ref_genrand_int32(C.c_int(n), _vp(out))
        return out

    def real(self, which, n):
        out = np.zeros(n)
        if self.kind == "o":
            f = self.lib.lto_genrand_real1 if which == 1 else self.lib.lto_genrand_real3
            for i in range(n):
                out[i] = f(C.byref(self._mt))
        else:
            f = self.lib.ref_genrand_real1 if which == 1 else self.lib.ref_genrand_real3
            f(C.c_int(n), _vp(out))
        return out

    def polintd(self, xa, ya, x):
        xa, ya = np.ascontiguousarray(xa, dtype=np.float64), np.ascontiguousarray(ya, dtype=np.float64)
        f = self.lib.lto_polintd if self.kind == "o" else self.lib.ref_polintd
        return f(_vp(xa), _vp(ya), C.c_double(x))

    def linint(self, xa, ya, x):
        xa, ya = np.ascontiguousarray(xa, dtype=np.float64), np.ascontiguousarray(ya, dtype=np.float64)
        y, m = C.c_double(), C.c_double()
        f = self.lib.lto_linint if self.kind == "o" else self.lib.ref_linint
        f(_vp(xa), _vp(ya), C.c_int(xa.size), C.c_double(x), C.byref(y), C.byref(m))
        return y.value, m.value

    def gridcell(self, ele_x, ele_y, x, y, checkele=0):
        """ele_x/ele_y: (4,elements). -> (P_ele, triangle) with P_ele=0 when nothing was set."""
        ex = np.asfortranarray(ele_x, dtype=np.float64)
        ey = np.asfortranarray(ele_y, dtype=np.float64)
        p, t = C.c_int(0), C.c_int(0)
        f = self.lib.lto_gridcell if self.kind == "o" else self.lib.ref_gridcell
        f(C.c_int(ex.shape[1]), _vp(ey), _vp(ex), C.c_double(x), C.c_double(y), C.byref(p),
          C.byref(t), C.c_int(checkele))
        return p.value, t.value

    def inpoly(self, x, y, e, onin=-1):
        e = np.asfortranarray(e, dtype=np.float64)
        if self.kind == "o":
            return self.lib.lto_inpoly(C.c_double(x), C.c_double(y), C.c_int(e.shape[0]), _vp(e), C.c_int(onin))
        return self.lib.ref_inpoly(C.c_double(x), C.c_double(y), C.c_int(e.shape[0]), _vp(e), C.c_int(onin))

    def tspsi(self, x, y):
        x, y = np.ascontiguousarray(x, dtype=np.float64), np.ascontiguousarray(y, dtype=np.float64)
        n = x.size
        yp, sig = np.zeros(n), np.zeros(n)
        ier, se = C.c_int(0), C.c_int(0)
        f = self.lib.lto_tspsi if self.kind == "o" else self.lib.ref_tspsi
        f(C.c_int(n), _vp(x), _vp(y), _vp(yp), _vp(sig), C.byref(ier), C.byref(se))
        return yp, sig, ier.value, se.value

    def hval(self, t, x, y, yp, sig, deriv=False):
        a = [np.ascontiguousarray(v, dtype=np.float64) for v in (x, y, yp, sig)]
        ier = C.c_int(0)
        name = ("lto_" if self.kind == "o" else "ref_") + ("hpval" if deriv else "hval")
        v = getattr(self.lib, name)(C.c_double(t), C.c_int(a[0].size), _vp(a[0]), _vp(a[1]),
                                    _vp(a[2]), _vp(a[3]), C.byref(ier))
        return v, ier.value

    def snhcsh(self, x):
        a, b, c = C.c_double(), C.c_double(), C.c_double()
        f = self.lib.lto_snhcsh if self.kind == "o" else self.lib.ref_snhcsh
        f(C.c_double(x), C.byref(a), C.byref(b), C.byref(c))
        return a.value, b.value, c.value


# ---------------------------------------------------------------------------------------------
# the CPU oracle behind the Python driver of the product (ltrans_amd.Engine): test infrastructure
import ltrans_amd  # noqa: E402


class OracleEngine(ltrans_amd.Engine):
    """oracle/libltrans_oracle.so (entry points lto_*) driven through the same methods as the HIP engine"""

    def __init__(self, cfg, lib=None, prefix="lto_"):
        self._open(cfg, lib if lib is not None else load_oracle(), prefix)


def engine(cfg, lib=None, prefix="lt_", **_):
    """Engine factory of the tests: the HIP engine, or -- for the keyword sets the test modules pass around
    as ``dict(lib=load_oracle(), prefix="lto_")`` -- the oracle"""
    if prefix == "lto_":
        return OracleEngine(cfg, lib, prefix)
    return ltrans_amd.Engine(cfg)


def create_bounds(lib, prefix, g, mask=None):
    """createBounds through the C ABI form shared by the oracle (lto_) and the engine's host side (lt_):
    -> dict like ltrans_amd.roms_synth.create_bounds"""
    import ctypes as C
    vi, uj = g["vi"], g["uj"]
    m = np.asfortranarray((g["mask_rho"] if mask is None else mask).astype(np.int32))
    F = lambda a: np.asfortranarray(a, dtype=np.float64)  # noqa: E731
    xu, yu, xv, yv = F(g["x_u"]), F(g["y_u"]), F(g["x_v"]), F(g["y_v"])
    cap = 2 * vi * uj
    nb, mb, mi, ni = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    bx, by, hx, hy = (np.zeros(cap) for _ in range(4))
    hid, land = np.zeros(cap, dtype=np.int32), np.zeros(cap, dtype=np.int32)
    bnd_x, bnd_y = np.zeros(2 * cap), np.zeros(2 * cap)
    f = getattr(lib, prefix + "create_bounds")
    f.restype = C.c_int
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    rc = f(C.c_int32(vi), C.c_int32(uj), vp(m), vp(xu), vp(yu), vp(xv), vp(yv), C.c_int32(cap), C.byref(nb), C.byref(mb),
           C.byref(mi), C.byref(ni), vp(bx), vp(by), vp(hx), vp(hy), vp(hid), vp(bnd_x), vp(bnd_y), vp(land))
    if rc != 0:
        raise RuntimeError(f"{prefix}create_bounds: rc = {rc}")
    nb, mb, mi, ni = nb.value, mb.value, mi.value, ni.value
    return dict(nbounds=nb, maxbound=mb, maxisland=mi, numislands=ni,
                bnd_x=np.asfortranarray(bnd_x[:2 * nb].reshape(2, nb, order="F")),
                bnd_y=np.asfortranarray(bnd_y[:2 * nb].reshape(2, nb, order="F")), land=land[:nb].copy(),
                bx=bx[:mb].copy(), by=by[:mb].copy(), hx=hx[:mi].copy(), hy=hy[:mi].copy(), hid=hid[:mi].copy())
