#!/usr/bin/env python3
"""Generate tests/golden/leaf_golden.npz from the REFERENCE's own Fortran.

Run in the build container (needs oracle/_ref/libltrans_ref.so = the reference's
gridcell / interpolation / point_in_polygon / random / tension modules compiled
with flang by oracle/Makefile).  The fixture holds inputs and the outputs the
reference produced; tests/test_golden.py replays the inputs through the CPU
oracle (and the GPU tests through the HIP kernels).  Data only -- no reference
source text is stored.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib  # noqa: E402


def spline_profiles(rng):
    """(x, y) profiles: random, monotone, convex, ties, wide dynamic range, float32
    values, zeros; N = 4 (WCTS_ITPI) and N = 84 / 124 (VTurb, ws = 21 / 31)."""
    out = []
    for trial in range(420):
        n = 4 if trial % 3 else (84 if trial % 2 else 124)
        x = np.cumsum(rng.uniform(0.05, 3.0, n)) - 50.0
        kind = trial % 7
        if kind == 0:
            y = rng.normal(0, 1, n)
        elif kind == 1:
            y = np.cumsum(rng.uniform(0, 1, n))
        elif kind == 2:
            y = (x - x[0]) ** 2 * rng.uniform(0.1, 2)
        elif kind == 3:
            y = np.round(rng.normal(0, 1, n), 1)
        elif kind == 4:
            y = np.exp(rng.normal(0, 3, n))
        elif kind == 5:
            y = np.float32(rng.normal(0, 0.3, n)).astype(np.float64)
        else:
            y = np.where(rng.random(n) < 0.3, 0.0, rng.normal(0, 1, n))
        out.append((x, y))
    # profiles known to exhaust the 10000-iteration Newton cap (SigErr /= 0): T ~ 2.026
    for (dx, s, d1, d2) in [(2.9553067065809557, 0.018950150622943956, 0.016022216157173513,
                             0.032467458443967104),
                            (2.970339478372427, 0.8239577552261154, 3.2958310209044615,
                             1.626474398537591)]:
        out.append(("raw", dx, s, d1, d2))
    return out


def main():
    R = oracle_lib.Leaf(oracle_lib.load_ref(), "r")
    rng = np.random.default_rng(20260101)
    G = {}
    # --- MT19937 (random_module.f90)
    for seed in (9, 5489, 4357):
        R.init_genrand(seed)
        G[f"mt_int32_seed{seed}"] = R.int32(1000)
        G[f"mt_real3_seed{seed}"] = R.real(3, 64)
        G[f"mt_real1_seed{seed}"] = R.real(1, 64)
    # --- polintd / linint (interpolation_module.f90)
    xa = np.array([[0., 3600., 7200.], [3600., 7200., 10800.], [-3600., 0., 3600.]])
    pol_in, pol_out = [], []
    for k in range(300):
        a = xa[k % 3]
        ya = rng.normal(0, 1, 3)
        x = a[1] + rng.uniform(-200, 3800)
        pol_in.append(np.concatenate([a, ya, [x]]))
        pol_out.append(R.polintd(a, ya, x))
    G["polintd_in"], G["polintd_out"] = np.array(pol_in), np.array(pol_out)
    lin_in, lin_out = [], []
    for k in range(200):
        n = 4 if k % 2 else 124
        x = np.cumsum(rng.uniform(0.1, 2, n))
        y = rng.normal(0, 1, n)
        t = rng.uniform(x[0], x[-1])
        lin_in.append((x, y, t))
        lin_out.append(R.linint(x, y, t))
    G["linint_n"] = np.array([len(a[0]) for a in lin_in])
    G["linint_x"] = np.concatenate([a[0] for a in lin_in])
    G["linint_y"] = np.concatenate([a[1] for a in lin_in])
    G["linint_t"] = np.array([a[2] for a in lin_in])
    G["linint_out"] = np.array(lin_out)
    # --- gridcell (gridcell_module.f90): a 3x3 patch of perturbed + one axis-aligned quad,
    # points random, on nodes, on edges, on horizontal edges, outside
    ex, ey = [], []
    for j in range(3):
        for i in range(3):
            px = lambda a, b: a * 10.0 + 0.7 * np.sin(a + 2 * b)   # noqa: E731
            py = lambda a, b: b * 10.0 + 0.7 * np.cos(2 * a - b)   # noqa: E731
            ex.append([px(i, j), px(i + 1, j), px(i + 1, j + 1), px(i, j + 1)])
            ey.append([py(i, j), py(i + 1, j), py(i + 1, j + 1), py(i, j + 1)])
    ex.append([40., 50., 50., 40.]); ey.append([0., 0., 10., 10.])      # axis aligned
    ex.append([50., 60., 60., 50.]); ey.append([0., 0., 10., 10.])
    ex, ey = np.array(ex).T.copy(order="F"), np.array(ey).T.copy(order="F")
    pts = [tuple(p) for p in rng.uniform(-2, 32, (300, 2))]
    pts += [(ex[k, e], ey[k, e]) for e in range(ex.shape[1]) for k in range(4)]          # nodes
    pts += [((ex[0, e] + ex[1, e]) / 2, (ey[0, e] + ey[1, e]) / 2) for e in range(ex.shape[1])]
    pts += [(45., 0.), (45., 10.), (40., 5.), (50., 5.), (50., 0.), (50., 10.), (55., 0.), (60., 5.),
            (45., 5.), (39.999, 5.), (50., 10.0000001), (47., 1e-300)]
    gc = []
    for (x, y) in pts:
        full = R.gridcell(ex, ey, x, y, 0)
        single = [R.gridcell(ex, ey, x, y, c)[1] for c in range(1, ex.shape[1] + 1)]
        gc.append([full[0], full[1]] + single)
    G["gc_ex"], G["gc_ey"], G["gc_pts"], G["gc_out"] = ex, ey, np.array(pts), np.array(gc)
    # --- inpoly (point_in_polygon_module.f90): closed polygons, generic + on-ray cases
    polys = []
    sq = np.array([[0, 0], [10, 0], [10, 10], [0, 10], [0, 0]], float)
    polys.append(sq)
    polys.append(np.array([[0, 0], [4, 0], [4, 4], [8, 4], [8, 0], [12, 0], [12, 8], [6, 12], [0, 8], [0, 0]], float))
    ang = np.linspace(0, 2 * np.pi, 33)
    star = np.c_[5 + (3 + 2 * np.cos(5 * ang)) * np.cos(ang), 5 + (3 + 2 * np.cos(5 * ang)) * np.sin(ang)]
    star[-1] = star[0]
    polys.append(star)
    ip_in, ip_out = [], []
    for pi, P in enumerate(polys):
        tests = [tuple(p) for p in rng.uniform(-2, 14, (150, 2))]
        tests += [tuple(v) for v in P[:-1]]                                   # on vertices
        tests += [(-1.0, v[1]) for v in P[:-1]] + [(v[0] - 0.5, v[1]) for v in P[:-1]]  # ray through vertices
        tests += [((P[k, 0] + P[k + 1, 0]) / 2, (P[k, 1] + P[k + 1, 1]) / 2) for k in range(len(P) - 1)]
        for (x, y) in tests:
            for onin in (-1, 0, 1):
                ip_in.append((pi, x, y, onin))
                ip_out.append(R.inpoly(x, y, P, onin))
    G["ip_npoly"] = np.array([len(P) for P in polys])
    G["ip_polys"] = np.concatenate(polys)
    G["ip_in"], G["ip_out"] = np.array(ip_in), np.array(ip_out)
    # --- TSPACK (tension_module.f90)
    sn_x = np.concatenate([rng.uniform(-0.5, 0.5, 50), rng.uniform(-40, 40, 50), [0.0, 0.5, -0.5, 0.5000001]])
    G["snhcsh_x"] = sn_x
    G["snhcsh_out"] = np.array([R.snhcsh(v) for v in sn_x])
    xs, ys, yps, sgs, meta, ev_t, ev_v = [], [], [], [], [], [], []
    for prof in spline_profiles(rng):
        if isinstance(prof[0], str):
            _, dx, s, d1, d2 = prof
            # 4 knots whose middle interval reproduces (s, d1, d2) closely enough to sit in the
            # slow-convergence band; outcome is whatever the reference does with it
            x = np.array([0.0, dx, 2 * dx, 3 * dx])
            y = np.array([0.0, (s - d1) * dx, (s - d1) * dx + s * dx, (s - d1) * dx + s * dx + (s + d2) * dx])
        else:
            x, y = prof
        yp, sg, ier, se = R.tspsi(x, y)
        n = x.size
        meta.append((n, ier, se))
        xs.append(x); ys.append(y); yps.append(yp); sgs.append(sg)
        ts = np.concatenate([[x[0] - 1.0, x[-1] + 1.0, x[1], x[-1]], rng.uniform(x[0], x[-1], 4)])
        for t in ts:
            hv, i1 = R.hval(t, x, y, yp, sg)
            hp, i2 = R.hval(t, x, y, yp, sg, deriv=True)
            ev_t.append(t); ev_v.append((hv, hp, i1, i2))
    G["sp_meta"] = np.array(meta)
    G["sp_x"], G["sp_y"] = np.concatenate(xs), np.concatenate(ys)
    G["sp_yp"], G["sp_sigma"] = np.concatenate(yps), np.concatenate(sgs)
    G["sp_eval_t"], G["sp_eval_out"] = np.array(ev_t), np.array(ev_v)
    out = os.path.join(HERE, "leaf_golden.npz")
    np.savez_compressed(out, **G)
    print("wrote", out, os.path.getsize(out), "bytes; SigErr profiles:", int((G["sp_meta"][:, 2] != 0).sum()))


if __name__ == "__main__":
    main()
