"""Init-time geometry, CPU side: the oracle's literal restatements of
initGrid's adjacency lists (hydrodynamic_module.f90:588-645), setEle_all (:1536-1676) and
createPolySpecs (settlement_module.f90:245-480) against the independent numpy constructions of
ltrans_amd.roms_synth, and their edge cases.  The device versions are held against these
restatements in tests/test_gpu_init_geometry.py."""
import numpy as np
import pytest

import cases
import ltrans_amd
import oracle_lib
from ltrans_amd import roms_synth as rs


@pytest.fixture(scope="module")
def okw():
    return dict(lib=oracle_lib.load_oracle(), prefix="lto_")


def _grid_engine(okw, name="tiny_settle", **grid_kw):
    case = cases.build_case(name)
    if grid_kw:
        spec = cases.CASES[name]
        vi, uj, us = spec["grid"]
        g = rs.make_grid(vi, uj, us, islands=spec["islands"], **grid_kw)
        case["g"], case["geo"] = g, rs.build_geometry(g)
    e = oracle_lib.engine(case["cfg"], **okw)
    e.set_grid(case["geo"])
    return case, e


def test_adjacency_matches_independent_construction(okw):
    case, e = _grid_engine(okw)
    geo, g = case["geo"], case["g"]
    for E, row, ref in ((geo["RE"], g["vi"], geo["r_Adjacent"]), (geo["UE"], g["ui"], geo["u_Adjacent"]),
                        (geo["VE"], g["vi"], geo["v_Adjacent"])):
        adj = e.build_adjacency(E, row)
        assert adj.shape == ref.shape
        assert np.array_equal(adj, ref)
        assert np.all(adj[:, 0] == np.arange(1, adj.shape[0] + 1)) and np.all(adj[:, 9] == 0)


def test_adjacency_window_is_the_references(okw):
    """The reference only looks at elements i-(row+2) .. i+row+2: with a row length that is too
    small the neighbours of the next grid row are not found (and a huge one finds them all)."""
    case, e = _grid_engine(okw)
    geo, g = case["geo"], case["g"]
    full = e.build_adjacency(geo["RE"], g["vi"])
    narrow = e.build_adjacency(geo["RE"], 0)                   # window i-2 .. i+2
    assert np.all(np.count_nonzero(narrow, axis=1) <= 3)
    assert np.array_equal(e.build_adjacency(geo["RE"], 10 * g["vi"]), full)


def test_locate_matches_independent_search(okw):
    case, e = _grid_engine(okw)
    g, geo = case["g"], case["geo"]
    rng = np.random.default_rng(3)
    x = rng.uniform(2.2 * g["dx"], (g["vi"] - 3.2) * g["dx"], 3000)
    y = rng.uniform(2.2 * g["dy"], (g["uj"] - 3.2) * g["dy"], 3000)
    keep = rs.locate_elements(g, geo, x, y, "r") > 0
    keep &= rs.locate_elements(g, geo, x, y, "u") > 0
    keep &= rs.locate_elements(g, geo, x, y, "v") > 0
    x, y = x[keep], y[keep]
    r, u, v, err, par = e.locate_particles(x, y)
    assert (err, par) == (0, 0)
    assert np.array_equal(r, rs.locate_elements(g, geo, x, y, "r"))
    assert np.array_equal(u, rs.locate_elements(g, geo, x, y, "u"))
    assert np.array_equal(v, rs.locate_elements(g, geo, x, y, "v"))


def test_locate_reports_the_first_particle_outside(okw):
    case, e = _grid_engine(okw)
    g = case["g"]
    x = np.full(6, 8.3 * g["dx"])
    y = np.full(6, 7.4 * g["dy"])
    x[4] = -5.0e6                                              # nowhere
    r, u, v, err, par = e.locate_particles(x, y)
    assert (err, par) == (1, 5)
    assert np.all(r[:4] > 0) and np.all(r[4:] == 0) and np.all(u[4:] == 0)
    r, u, v, err, par = e.locate_particles(x[4:], y[4:])       # the first particle: all three grids tried
    assert (err, par) == (3, 1)


def test_locate_on_shared_nodes_depends_on_the_previous_particle(okw):
    """A point on a node shared by four elements is 'in' all of them: setEle_all takes the first of
    the previous particle's adjacency row that contains it, so the answer depends on where the
    previous particle was (rectilinear grid, so that nodes and edges can be hit exactly)."""
    case, e = _grid_engine(okw, curvilinear=0.0)
    g, geo = case["g"], case["geo"]
    xn, yn = float(g["x_rho"][6, 5]), float(g["y_rho"][6, 5])
    dx, dy = g["dx"], g["dy"]
    seen = set()
    for ox, oy in ((-0.5, -0.5), (0.5, -0.5), (-0.5, 0.5), (0.5, 0.5)):
        r, u, v, err, par = e.locate_particles(np.array([xn + ox * dx, xn]), np.array([yn + oy * dy, yn]))
        assert err == 0
        assert r[1] == r[0]                                    # the previous particle's own element comes first
        seen.add(int(r[1]))
    assert len(seen) == 4
    # as the first particle it is the scan of all elements that decides
    r, u, v, err, par = e.locate_particles(np.array([xn]), np.array([yn]))
    assert err == 0 and int(r[0]) == min(seen)


def test_polyspecs_and_lists(okw):
    case, e = _grid_engine(okw)
    hab = case["habitat"]
    out = e.create_polyspecs(hab)
    assert np.array_equal(out["polyspecs"], hab["polyspecs"])
    assert np.array_equal(out["holespecs"], hab["holespecs"])
    assert np.array_equal(out["polyholes_ptr"], hab["polyholes_ptr"])
    nph = int(hab["polyholes_ptr"][-1])
    assert np.array_equal(out["polyholes_ids"][:nph], hab["polyholes_ids"][:nph])
    # every listed polygon overlaps the element's bounding box (roms_synth lists that superset) ...
    ptr, ids = out["elepolys_ptr"], out["elepolys_ids"]
    sptr, sids = hab["elepolys_ptr"], hab["elepolys_ids"]
    assert ptr[-1] > 0
    for i in range(ptr.size - 1):
        mine, sup = list(ids[ptr[i]:ptr[i + 1]]), set(sids[sptr[i]:sptr[i + 1]])
        assert len(set(mine)) == len(mine) and mine == sorted(mine)
        assert set(mine) <= sup
    # ... and an element with a polygon vertex inside it lists that polygon
    g, geo = case["g"], case["geo"]
    polys = hab["polys"]
    ele = rs.locate_elements(g, geo, polys[:, 3].copy(), polys[:, 4].copy(), "r")
    for row, el in enumerate(ele):
        if el > 0:
            assert int(round(polys[row, 0])) in ids[ptr[el - 1]:ptr[el]]


def test_polyspecs_rejects_ids_outside_the_declared_range(okw):
    case, e = _grid_engine(okw)
    hab = dict(case["habitat"])
    hab["maxpolyid"] = hab["minpolyid"]                        # three polygons, range of one
    with pytest.raises(ltrans_amd.LtError):
        e.create_polyspecs(hab)


def test_adjacency_overflow_is_reported(okw):
    """eleven elements that all share node 1: the reference would write r_Adjacent(i,11)"""
    case, e = _grid_engine(okw)
    E = np.ones((4, 11), dtype=np.int32, order="F")
    E[1:, :] = np.arange(2, 35, dtype=np.int32).reshape(3, 11)
    with pytest.raises(ltrans_amd.LtError):
        e.build_adjacency(E, 20)
