"""Live cross-check of the CPU oracle against the reference's own modules
compiled with flang (oracle/_ref): fresh random inputs every run section.
Skipped when oracle/_ref was not built (no /root/reference or no flang)."""
import numpy as np
import pytest

import oracle_lib


@pytest.fixture(scope="module")
def pair(oracle, reflib):
    return oracle_lib.Leaf(oracle, "o"), oracle_lib.Leaf(reflib, "r")


def test_mt_long_stream(pair):
    O, R = pair
    for seed in (1, 9, 123456789):
        O.init_genrand(seed)
        R.init_genrand(seed)
        assert np.array_equal(O.int32(5000), R.int32(5000))


def test_tspack_random(pair):
    O, R = pair
    rng = np.random.default_rng(7)
    for trial in range(3000):
        n = 4 if trial % 4 else int(rng.integers(5, 130))
        x = np.cumsum(rng.uniform(0.05, 3.0, n)) - 50.0
        y = [rng.normal(0, 1, n), np.cumsum(rng.uniform(0, 1, n)), np.round(rng.normal(0, 1, n), 1),
             np.exp(rng.normal(0, 3, n))][trial % 4]
        a, b = O.tspsi(x, y), R.tspsi(x, y)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1][:n - 1], b[1][:n - 1])
        assert a[2:] == b[2:]
        for t in (x[0] - 1, x[-1] + 1, *rng.uniform(x[0], x[-1], 2)):
            assert O.hval(t, x, y, a[0], a[1]) == R.hval(t, x, y, b[0], b[1])
            assert O.hval(t, x, y, a[0], a[1], True) == R.hval(t, x, y, b[0], b[1], True)


def test_sigs_small_sigma_secant_branch(pair):
    """profiles built to enter the monotonicity secant loop, including iterates with
    SIG <= .5 where the F90 text leaves CONT unset (oracle pins it .TRUE.)."""
    O, R = pair
    rng = np.random.default_rng(11)
    hits = 0
    for _ in range(20000):
        dx = rng.uniform(0.5, 2.0, 3)
        x = np.concatenate([[0.0], np.cumsum(dx)])
        s = rng.uniform(0.5, 2.0)
        # data increasing with an inflection: D1*D2 < 0 and slopes of one sign
        y = np.array([0.0, s * dx[0] * rng.uniform(0.01, 0.3), 0, 0])
        y[2] = y[1] + s * dx[1]
        y[3] = y[2] + s * dx[2] * rng.uniform(0.01, 0.3)
        a, b = O.tspsi(x, y), R.tspsi(x, y)
        assert np.array_equal(a[1][:3], b[1][:3]) and a[2:] == b[2:]
        hits += int(((a[1][:3] > 0) & (a[1][:3] < 85)).any())
    assert hits > 100


def test_gridcell_random(pair):
    O, R = pair
    rng = np.random.default_rng(3)
    for _ in range(300):
        c = rng.uniform(-5, 5, 2)
        q = np.array([[0, 0], [10, 0], [10, 10], [0, 10]], float) + rng.uniform(-2, 2, (4, 2)) + c
        ex, ey = q[:, :1].copy(order="F"), q[:, 1:].copy(order="F")
        for _ in range(20):
            x, y = rng.uniform(-8, 18, 2) + c
            assert O.gridcell(ex, ey, x, y, 1) == R.gridcell(ex, ey, x, y, 1)


def test_inpoly_random(pair):
    O, R = pair
    rng = np.random.default_rng(5)
    for _ in range(200):
        n = int(rng.integers(3, 40))
        ang = np.sort(rng.uniform(0, 2 * np.pi, n))
        rad = rng.uniform(2, 10, n)
        P = np.c_[rad * np.cos(ang), rad * np.sin(ang)]
        P = np.vstack([P, P[:1]])
        if rng.random() < 0.5:
            P = np.round(P)          # integer vertices: rays through vertices become likely
        for _ in range(30):
            x, y = (np.round(rng.uniform(-11, 11, 2)) if rng.random() < 0.5 else rng.uniform(-11, 11, 2))
            for onin in (-1, 0, 1):
                assert O.inpoly(x, y, P, onin) == R.inpoly(x, y, P, onin)


def test_timing_loops_of_port_and_reference_compute_the_same(pair):
    """oracle/time_leaves.py compares the cost of the port with the reference's: the two loops must do the same work
    -- their accumulated results are identical, bit for bit"""
    import ctypes as C
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import time_leaves as tl
    ora, ref = C.CDLL(tl.ORACLE_SO), C.CDLL(tl.REF_SO)
    inp = tl._inputs()
    vp = tl._vp

    def acc_of(lib, prefix, name, args):
        a = C.c_double()
        getattr(lib, prefix + name)(*args, C.byref(a))
        return a.value

    for n in (4, 124):
        x, y, t, neval = inp[n]
        args = (C.c_int(3000), C.c_int(n), C.c_int(x.shape[0]), vp(x), vp(y), vp(t), C.c_int(neval))
        assert acc_of(ora, "lto_time_", "tspsi_hval", args) == acc_of(ref, "ref_time_", "tspsi_hval", args)
    ex, ey, px, py, ne = inp["gridcell"]
    args = (C.c_int(20000), C.c_int(ne), vp(ey), vp(ex), C.c_int(px.size), vp(px), vp(py))
    a, b = acc_of(ora, "lto_time_", "gridcell", args), acc_of(ref, "ref_time_", "gridcell", args)
    assert a == b and a > 100
    e, qx, qy = inp["inpoly"]
    args = (C.c_int(4000), C.c_int(e.shape[0]), vp(e), C.c_int(qx.size), vp(qx), vp(qy))
    a, b = acc_of(ora, "lto_time_", "inpoly", args), acc_of(ref, "ref_time_", "inpoly", args)
    assert a == b and 100 < a < 3900
