"""The slab index of the point-in-polygon tests (csrc/lt_device.h: PipIndex; mbounds / ibounds of boundary_module.f90:
1515-1614 through it) against the reference's own algorithm -- the oracle's inpoly, pinned to the compiled
point_in_polygon_module (tests/test_oracle_ref.py) -- WITHOUT a GPU: the test build runs the index builder of
lt_set_bounds and the kernels' query functions on the host (lt_test_pip_index_host).  Polygons as createBounds makes
them: runs of edges along x at one ordinate, runs along y, staircases, islands; points on vertices' ordinates (the
hand-over to the full algorithm), on edges, next to long horizontal runs."""
import ctypes as C

import numpy as np
import pytest

import hip_test_lib


def _staircase_polygon(rng, nx, ny, jitter):
    """a closed polygon along cell edges of an nx x ny block of cells with a ragged rim (what a land mask gives)"""
    south = rng.integers(0, 3, nx); north = ny - rng.integers(0, 3, nx)
    pts = []
    for i in range(nx):                      # south rim, west -> east
        pts += [(i, south[i]), (i + 1, south[i])]
    for i in range(nx - 1, -1, -1):          # north rim, east -> west
        pts += [(i + 1, north[i]), (i, north[i])]
    out = [pts[0]]
    for p in pts[1:] + [pts[0]]:
        if p != out[-1]:
            out.append(p)
    P = np.array(out, dtype=np.float64) * 1000.0
    if jitter:
        P[:-1] += rng.uniform(-jitter, jitter, (len(P) - 1, 2))
        P[-1] = P[0]
    return P


def _query(lib, P, pid, q):
    o_in = np.zeros(len(q), dtype=np.int32); o_slow = np.zeros(len(q), dtype=np.int32)
    px, py = np.ascontiguousarray(P[:, 0]), np.ascontiguousarray(P[:, 1])
    qx, qy = np.ascontiguousarray(q[:, 0]), np.ascontiguousarray(q[:, 1])
    vp = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    lib.lt_test_pip_index_host.restype = C.c_int
    rc = lib.lt_test_pip_index_host(vp(px), vp(py), vp(pid) if pid is not None else None, C.c_int32(len(px)), vp(qx), vp(qy),
                                    C.c_int32(len(q)), vp(o_in), vp(o_slow))
    assert rc >= 0, rc
    return o_in, o_slow, rc


def _reference(oracle, P, q):
    """the oracle's inpoly (point_in_polygon_module.f90:25-167; a point on the boundary counts as inside, as mbounds /
    ibounds call it)"""
    E = np.asfortranarray(P, dtype=np.float64)
    oracle.lto_inpoly.restype = C.c_int
    return np.array([oracle.lto_inpoly(C.c_double(x), C.c_double(y), C.c_int(len(P)), E.ctypes.data_as(C.c_void_p), C.c_int(-1))
                     for x, y in q], dtype=np.int32)


def _points(rng, P, n):
    lo, hi = P.min(0) - 500.0, P.max(0) + 500.0
    q = rng.uniform(lo, hi, (n, 2))
    k = n // 4
    v = P[rng.integers(0, len(P), k)]
    q[:k, 1] = v[:, 1]                                             # on a vertex's ordinate: the hand-over
    q[k:2 * k] = P[rng.integers(0, len(P), k)] + rng.uniform(-3.0, 3.0, (k, 2))   # next to vertices
    e = rng.integers(0, len(P) - 1, k)
    t = rng.uniform(0, 1, (k, 1))
    q[2 * k:3 * k] = P[e] * (1 - t) + P[e + 1] * t                  # on (or a rounding away from) edges
    return q


@pytest.mark.parametrize("seed,jitter", [(1, 0.0), (2, 0.0), (3, 40.0), (4, 2.0)])
def test_one_polygon_against_the_reference_algorithm(oracle, seed, jitter):
    lib = hip_test_lib.load()
    rng = np.random.default_rng(seed)
    P = _staircase_polygon(rng, 60, 25, jitter)
    q = _points(rng, P, 6000)
    o_in, o_slow, entries = _query(lib, P, None, q)
    want = _reference(oracle, P, q)
    dec = o_slow == 0
    assert dec.sum() > 3000 and (~dec).sum() > 100
    assert np.array_equal(o_in[dec], (want[dec] != 0).astype(np.int32))
    # the lists hold what a ray can cross and nothing else: a handful of entries per slab although the rim's runs
    # put tens of edges on one ordinate
    assert entries < 40 * len(P)


def test_islands_inside_any(oracle):
    lib = hip_test_lib.load()
    rng = np.random.default_rng(9)
    polys, pid = [], []
    for k, (ox, oy) in enumerate([(0.0, 0.0), (30000.0, 4000.0), (12000.0, 30000.0)]):
        Pk = _staircase_polygon(rng, 8 + 3 * k, 6 + k, 0.0) + np.array([ox, oy])
        polys.append(Pk); pid += [1001 + k] * len(Pk)
    P = np.vstack(polys); pid = np.array(pid, dtype=np.int32)
    q = _points(rng, P, 5000)
    o_in, o_slow, _ = _query(lib, P, pid, q)
    want = np.zeros(len(q), dtype=np.int32)
    for Pk in polys:
        want |= (_reference(oracle, Pk, q) != 0).astype(np.int32)
    dec = o_slow == 0
    assert dec.sum() > 2500
    assert np.array_equal(o_in[dec], want[dec])
