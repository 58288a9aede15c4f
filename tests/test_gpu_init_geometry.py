"""Init-time geometry on the device (ltransv.2b_amd/csrc/lt_init.h) against the
oracle's literal restatements of initGrid's adjacency loops (hydrodynamic_module.f90:588-645),
setEle_all (:1536-1676) and createPolySpecs (settlement_module.f90:245-480): identical integers,
through the C ABI (lt_build_adjacency, lt_locate_particles, lt_create_polyspecs)."""
import time

import numpy as np
import pytest

import cases
import ltrans_amd
import oracle_lib
from ltrans_amd import roms_synth as rs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def okw():
    return dict(lib=oracle_lib.load_oracle(), prefix="lto_")


def _engines(okw, vi, uj, islands, curvilinear=0.08, settle=False, polygons=(), holes=()):
    g = rs.make_grid(vi, uj, 8, islands=islands, curvilinear=curvilinear)
    geo = rs.build_geometry(g)
    cfg = ltrans_amd.LtConfig(numpar=64, us=8, ws=9, dt=3600, idt=120, settlementon=1 if settle else 0,
                              holesExist=1 if holes else 0)
    hip, ora = ltrans_amd.Engine(cfg), oracle_lib.engine(cfg, **okw)
    hip.set_grid(geo)
    ora.set_grid(geo)
    hab = rs.make_habitat(g, geo, list(polygons), list(holes)) if polygons else None
    return g, geo, hip, ora, hab


@pytest.mark.parametrize("islands", [[], [(10, 14, 6, 9), (30, 33, 20, 28)]])
def test_adjacency(okw, islands):
    g, geo, hip, ora, _ = _engines(okw, 48, 40, islands)
    for E, row in ((geo["RE"], g["vi"]), (geo["UE"], g["ui"]), (geo["VE"], g["vi"])):
        a, b = hip.build_adjacency(E, row), ora.build_adjacency(E, row)
        assert np.array_equal(a, b)
    # the reference's window: a wrong row length gives the reference's (incomplete) lists too
    assert np.array_equal(hip.build_adjacency(geo["RE"], 3), ora.build_adjacency(geo["RE"], 3))


def _points(g, rng, n, exact):
    """random interior points; with exact=True (rectilinear grid) a third of them on rho / u / v
    nodes and on element edges, where several elements contain the point"""
    x = rng.uniform(1.2 * g["dx"], (g["vi"] - 2.2) * g["dx"], n)
    y = rng.uniform(1.2 * g["dy"], (g["uj"] - 2.2) * g["dy"], n)
    if exact:
        k = n // 9
        for s, (X, Y) in enumerate(((g["x_rho"], g["y_rho"]), (g["x_u"], g["y_u"]), (g["x_v"], g["y_v"]))):
            i = rng.integers(1, X.shape[0] - 1, k)
            j = rng.integers(1, X.shape[1] - 1, k)
            sl = slice(3 * s * k, (3 * s + 1) * k)
            x[sl], y[sl] = X[i, j], Y[i, j]                                        # on a node
            sl = slice((3 * s + 1) * k, (3 * s + 2) * k)
            x[sl], y[sl] = X[i, j] + rng.uniform(0.05, 0.95, k) * g["dx"], Y[i, j]  # on a horizontal edge
            sl = slice((3 * s + 2) * k, (3 * s + 3) * k)
            x[sl], y[sl] = X[i, j], Y[i, j] + rng.uniform(0.05, 0.95, k) * g["dy"]  # on a vertical edge
        p = rng.permutation(n)
        x, y = x[p], y[p]
    return x, y


@pytest.mark.parametrize("curv,exact", [(0.08, False), (0.0, True)])
def test_locate_particles(okw, curv, exact):
    g, geo, hip, ora, _ = _engines(okw, 40, 32, [(10, 14, 6, 9)], curvilinear=curv)
    rng = np.random.default_rng(5)
    x, y = _points(g, rng, 9000, exact)
    # keep only points that are in some element of all three grids, so the whole array is searched
    keep = np.ones(x.size, dtype=bool)
    for kind in "ruv":
        keep &= rs.locate_elements(g, geo, x, y, kind) > 0
    x, y = x[keep].copy(), y[keep].copy()
    r, u, v, err, par = ora.locate_particles(x, y)
    while err != 0:
        x, y = np.delete(x, par - 1), np.delete(y, par - 1)
        r, u, v, err, par = ora.locate_particles(x, y)
    assert x.size > 8000
    hr, hu, hv, herr, hpar = hip.locate_particles(x, y)
    assert (herr, hpar) == (0, 0)
    assert np.array_equal(hr, r) and np.array_equal(hu, u) and np.array_equal(hv, v)
    if exact:   # the order dependence is really exercised: reversing the particles changes some answers
        rr, _, _, e2, _ = ora.locate_particles(x[::-1].copy(), y[::-1].copy())
        assert e2 == 0 and np.count_nonzero(rr[::-1] != r) > 0
        hrr, _, _, _, _ = hip.locate_particles(x[::-1].copy(), y[::-1].copy())
        assert np.array_equal(hrr, rr)


def test_locate_particles_error_codes(okw):
    g, geo, hip, ora, _ = _engines(okw, 40, 32, [(10, 14, 6, 9)])
    rng = np.random.default_rng(6)
    x, y = _points(g, rng, 500, False)
    x[200] = 12.0 * g["dx"]                  # inside the island block: no element of any grid
    y[200] = 7.5 * g["dy"]
    x[350] = -1.0e7
    a, b = hip.locate_particles(x, y), ora.locate_particles(x, y)
    assert a[3:] == b[3:] and b[3] != 0
    for k in range(3):
        assert np.array_equal(a[k], b[k])
    a, b = hip.locate_particles(x[350:], y[350:]), ora.locate_particles(x[350:], y[350:])
    assert a[3:] == b[3:] == (3, 1)
    # u / v grids end half a cell before the rho grid: a point in a rho element but in no u element
    x2 = np.array([20.3 * g["dx"], 0.2 * g["dx"]])
    y2 = np.array([15.3 * g["dy"], 15.3 * g["dy"]])
    a, b = hip.locate_particles(x2, y2), ora.locate_particles(x2, y2)
    assert a[3:] == b[3:]
    for k in range(3):
        assert np.array_equal(a[k], b[k])


def test_create_polyspecs(okw):
    polygons = [(6000.0, 5000.0, 2500.0, 12), (16000.0, 9000.0, 3000.0, 9), (18000.0, 4000.0, 1500.0, 7),
                (30000.0, 20000.0, 400.0, 5),       # smaller than an element: found through its vertices
                (25000.0, 12000.0, 9000.0, 4)]      # a square covering whole elements: found through inpoly
    holes = [(0, 6200.0, 5200.0, 900.0, 8), (4, 25000.0, 12000.0, 2000.0, 6), (4, 27000.0, 14000.0, 500.0, 5)]
    g, geo, hip, ora, hab = _engines(okw, 40, 32, [(10, 14, 6, 9)], settle=True, polygons=polygons, holes=holes)
    a, b = hip.create_polyspecs(hab), ora.create_polyspecs(hab)
    for k in ("polyspecs", "holespecs", "elepolys_ptr", "polyholes_ptr"):
        assert np.array_equal(a[k], b[k]), k
    n, m = int(b["elepolys_ptr"][-1]), int(b["polyholes_ptr"][-1])
    assert n > 50 and m == 3
    assert np.array_equal(a["elepolys_ids"][:n], b["elepolys_ids"][:n])
    assert np.array_equal(a["polyholes_ids"][:m], b["polyholes_ids"][:m])
    # elements wholly inside the big square list it although none of its 5 vertices is in them
    ptr, ids = a["elepolys_ptr"], a["elepolys_ids"]
    big = hab["minpolyid"] + 4
    inside = rs.locate_elements(g, geo, np.array([25000.0]), np.array([12000.0]), "r")[0]
    assert big in ids[ptr[inside - 1]:ptr[inside]]


def test_create_polyspecs_without_holes_and_empty_locate(okw):
    polygons = [(6000.0, 5000.0, 2500.0, 12), (30000.0, 20000.0, 400.0, 5)]
    g, geo, hip, ora, hab = _engines(okw, 40, 32, [(10, 14, 6, 9)], settle=True, polygons=polygons)
    assert hab["hedges"] == 0
    a, b = hip.create_polyspecs(hab), ora.create_polyspecs(hab)
    for k in ("polyspecs", "elepolys_ptr", "polyholes_ptr"):
        assert np.array_equal(a[k], b[k]), k
    n = int(b["elepolys_ptr"][-1])
    assert n > 0 and np.array_equal(a["elepolys_ids"][:n], b["elepolys_ids"][:n])
    assert int(a["polyholes_ptr"][-1]) == 0
    for e in (hip, ora):
        r, u, v, err, par = e.locate_particles(np.zeros(0), np.zeros(0))
        assert r.size == 0 and (err, par) == (0, 0)


def test_settlement_with_device_built_lists(okw):
    """tiny_settle with the habitat installed by lt_create_polyspecs instead of lt_set_habitat:
    same trajectories and settlement as the oracle with its own createPolySpecs."""
    res = []
    for kw in ({}, okw):
        case = cases.build_case("tiny_settle")
        case["habitat_raw"], case["habitat"] = case["habitat"], None
        res.append(cases.run_case(case, ltrans_amd.Engine, kw, n_external=2))
    a, b = res
    assert cases.rel_error(a["par"], b["par"]) <= 1e-9
    for k in ("settle", "endpoly", "dead"):
        assert np.array_equal(a[k], b[k])
    assert a["settle"].sum() > 0


def test_locate_full_size(okw):
    """1M particles on the 512x256 grid (BASELINE configs[1] shape): agrees with the independent
    host search of roms_synth and takes well under a second."""
    g = rs.make_grid(512, 256, 8)
    geo = rs.build_geometry(g)
    cfg = ltrans_amd.LtConfig(numpar=64, us=8, ws=9, dt=3600, idt=120)
    hip = ltrans_amd.Engine(cfg)
    hip.set_grid(geo)
    rng = np.random.default_rng(8)
    n = 1_000_000
    x = rng.uniform(2.5 * g["dx"], (g["vi"] - 3.5) * g["dx"], n)
    y = rng.uniform(2.5 * g["dy"], (g["uj"] - 3.5) * g["dy"], n)
    hip.locate_particles(x[:1000], y[:1000])
    t0 = time.perf_counter()
    r, u, v, err, par = hip.locate_particles(x, y)
    dt = time.perf_counter() - t0
    assert (err, par) == (0, 0)
    for kind, got in (("r", r), ("u", u), ("v", v)):
        assert np.array_equal(got, rs.locate_elements(g, geo, x, y, kind))
    t0 = time.perf_counter()
    for E, row in ((geo["RE"], g["vi"]), (geo["UE"], g["ui"]), (geo["VE"], g["vi"])):
        adj = hip.build_adjacency(E, row)
    dt_adj = time.perf_counter() - t0
    assert np.array_equal(adj, geo["v_Adjacent"])
    print(f"lt_locate_particles: 1M particles x 3 grids in {dt * 1e3:.0f} ms (host side included); "
          f"lt_build_adjacency x 3 grids of ~130k elements in {dt_adj * 1e3:.0f} ms")
    assert dt < 5.0


def test_adjacency_overflow_and_list_capacity(okw):
    import ctypes as C
    g, geo, hip, ora, hab = _engines(okw, 40, 32, [(10, 14, 6, 9)], settle=True,
                                     polygons=[(6000.0, 5000.0, 2500.0, 12)])
    E = np.ones((4, 11), dtype=np.int32, order="F")
    E[1:, :] = np.arange(2, 35, dtype=np.int32).reshape(3, 11)
    with pytest.raises(ltrans_amd.LtError):
        hip.build_adjacency(E, 20)
    # lt_get_polyspecs with id arrays that are too small: an error, and the ptr arrays tell the size
    hip.create_polyspecs(hab)
    ptr = np.zeros(geo["rho_elements"] + 1, dtype=np.int32)
    ids = np.zeros(1, dtype=np.int32)
    f = hip.lib.lt_get_polyspecs
    f.restype = C.c_int
    rc = f(hip._h, None, None, ptr.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), C.c_int32(1), None, None,
           C.c_int32(0))
    assert rc == -1 and ptr[-1] > 1
