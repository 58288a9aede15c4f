"""TEST INFRASTRUCTURE.  What the CPU port (oracle/ltrans_oracle.c) costs against the reference's own Fortran where
the reference compiles here (oracle/_ref: the five leaf modules, flang -O2): the same inputs through the same loops
(lto_time_* in the oracle, ref_time_* in oracle/ref_shim.f90), one thread.

  TSPSI(N=4) + HVAL              the fit of find_currents / WCTS_ITPI (tension_module.f90:231, 980): 36 per particle-step
  TSPSI(N=124) + 60 x (HPVAL+HVAL)   VTurb's profile and its walk (ver_turb_module.f90:296-345) at ws = 31, idt = 120
  gridcell, one element          the probe of setEle (gridcell_module.f90:26)
  inpoly, 1537 points            mbounds (point_in_polygon_module.f90:25) on a boundary of the bench grid's size

    python oracle/time_leaves.py        -> one JSON object; bench.py's cpu_baseline leg embeds `ratio`
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libltrans_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libltrans_ref.so")


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _inputs():
    rng = np.random.default_rng(4)
    nprof = 1024
    out = {}
    for n, neval in ((4, 1), (124, 60)):
        # level depths (increasing, uneven), a smooth profile with noise: some intervals need SIGS' iterations
        x = np.cumsum(rng.uniform(0.3, 3.0, (nprof, n)), axis=1) - 40.0
        y = 0.2 * np.sin(x / 7.0 + rng.uniform(0, 6, (nprof, 1))) + 0.02 * rng.standard_normal((nprof, n))
        t = x[:, :1] + (x[:, -1:] - x[:, :1]) * rng.uniform(0.02, 0.98, (nprof, neval))
        out[n] = (np.ascontiguousarray(x), np.ascontiguousarray(y), np.ascontiguousarray(t), neval)
    # 4096 quadrilateral elements of a gently curved grid; points inside or next to the element tested
    ne = 4096
    i, j = np.meshgrid(np.arange(64), np.arange(64), indexing="ij")
    cx, cy = (i * 1000.0 + 80.0 * np.sin(j / 9.0)).ravel(), (j * 1000.0 + 80.0 * np.cos(i / 7.0)).ravel()
    ex = np.stack([cx, cx + 1000.0, cx + 1010.0, cx + 5.0])       # (4, elements) column-major = Fortran ele_x(4,elements)
    ey = np.stack([cy, cy + 8.0, cy + 1003.0, cy + 1000.0])
    px = cx + rng.uniform(-200.0, 1200.0, ne)
    py = cy + rng.uniform(-200.0, 1200.0, ne)
    out["gridcell"] = (np.asfortranarray(ex), np.asfortranarray(ey), np.ascontiguousarray(px), np.ascontiguousarray(py), ne)
    # a closed boundary of 1537 points (512x256 grid: 2 x (512 + 256) + 1), points inside its bounding box
    k = 1536
    th = np.linspace(0.0, 2.0 * np.pi, k, endpoint=False)
    r = 1.0e5 * (1.0 + 0.15 * np.sin(5 * th) + 0.02 * rng.standard_normal(k))
    bx, by = 2.5e5 + 2.0 * r * np.cos(th), 1.3e5 + r * np.sin(th)
    e = np.asfortranarray(np.stack([np.append(bx, bx[0]), np.append(by, by[0])], axis=1))     # e(n,2)
    qx = rng.uniform(bx.min(), bx.max(), 2048)
    qy = rng.uniform(by.min(), by.max(), 2048)
    out["inpoly"] = (e, np.ascontiguousarray(qx), np.ascontiguousarray(qy))
    return out


def _time(fn, budget_s):
    """ns per repetition: repetitions doubled until a call takes budget_s / 4, best of three at that count"""
    nrep = 256
    while True:
        t0 = time.perf_counter()
        fn(nrep)
        dt = time.perf_counter() - t0
        if dt > budget_s / 4 or nrep > (1 << 28):
            break
        nrep *= 2
    best = dt
    for _ in range(2):
        t0 = time.perf_counter()
        fn(nrep)
        best = min(best, time.perf_counter() - t0)
    return best / nrep * 1e9


def measure(budget_s=0.4):
    """{case: {"port_ns", "reference_ns", "ratio"}} + "advection_step" (36 fits, one inpoly, four probes), or None when
    the reference build is absent"""
    if not (os.path.exists(REF_SO) and os.path.exists(ORACLE_SO)):
        return None
    ora, ref = C.CDLL(ORACLE_SO), C.CDLL(REF_SO)
    inp = _inputs()
    acc = C.c_double()
    res = {}

    def both(name, call):
        ns = {}
        for tag, lib, prefix in (("port", ora, "lto_time_"), ("reference", ref, "ref_time_")):
            ns[tag] = _time(lambda nrep: call(getattr(lib, prefix + name.split(":")[0]), nrep), budget_s)
        res[name.split(":")[-1]] = {"port_ns": round(ns["port"], 1), "reference_ns": round(ns["reference"], 1),
                                    "ratio": round(ns["port"] / ns["reference"], 3)}

    for n, label in ((4, "tspsi4_hval"), (124, "tspsi124_60x_hpval_hval")):
        x, y, t, neval = inp[n]
        both("tspsi_hval:" + label, lambda f, nrep: f(C.c_int(nrep), C.c_int(n), C.c_int(x.shape[0]), _vp(x), _vp(y), _vp(t),
                                                       C.c_int(neval), C.byref(acc)))
    ex, ey, px, py, ne = inp["gridcell"]
    both("gridcell", lambda f, nrep: f(C.c_int(nrep), C.c_int(ne), _vp(ey), _vp(ex), C.c_int(px.size), _vp(px), _vp(py),
                                       C.byref(acc)))
    e, qx, qy = inp["inpoly"]
    both("inpoly:inpoly_1537", lambda f, nrep: f(C.c_int(nrep), C.c_int(e.shape[0]), _vp(e), C.c_int(qx.size), _vp(qx), _vp(qy),
                                                 C.byref(acc)))
    w = {"tspsi4_hval": 36, "inpoly_1537": 1, "gridcell": 4}
    port = sum(w[k] * res[k]["port_ns"] for k in w)
    refc = sum(w[k] * res[k]["reference_ns"] for k in w)
    res["advection_step"] = {"weights": w, "port_ns": round(port, 1), "reference_ns": round(refc, 1), "ratio": round(port / refc, 3)}
    return res


if __name__ == "__main__":
    r = measure(float(sys.argv[1]) if len(sys.argv) > 1 else 0.4)
    print(json.dumps(r, indent=1))
