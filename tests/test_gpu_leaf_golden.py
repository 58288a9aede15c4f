"""The DEVICE leaf numerics (lt_math.h: polintd, gridcell, SNHCSH, YPC1 / SIGS / HVAL / HPVAL in
the reference-shaped form and in the forms k_rk's fast path uses) against the vectors the
REFERENCE's own Fortran modules produced (tests/golden/leaf_golden.npz, made with the flang
build of gridcell / interpolation / tension modules): the parity that is pinned to the
reference, checked on the GPU itself and not only through the CPU oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import ltrans_amd

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "leaf_golden.npz"))


@pytest.fixture(scope="module")
def leaf():
    lib = ltrans_amd.load_hip_library()
    lib.lt_test_leaf.restype = C.c_int

    def run(kind, rows, n_out):
        a = np.ascontiguousarray(rows, dtype=np.float64)
        n = a.shape[0]
        a = a.reshape(n, -1)
        out = np.zeros((n, n_out), dtype=np.float64)
        rc = lib.lt_test_leaf(C.c_int32(kind), a.ctypes.data_as(C.c_void_p), C.c_int32(n), C.c_int32(a.shape[1]),
                              out.ctypes.data_as(C.c_void_p), C.c_int32(n_out))
        assert rc == 0
        return out
    return run


def test_polintd_bit_exact(leaf):
    got = leaf(1, G["polintd_in"], 1)[:, 0]
    assert np.array_equal(got, G["polintd_out"])


def test_gridcell_every_branch(leaf):
    ex, ey, pts, out = G["gc_ex"], G["gc_ey"], G["gc_pts"], G["gc_out"]
    rows, exp = [], []
    for p, o in zip(pts, out):
        for e in range(ex.shape[1]):
            rows.append(np.concatenate([ex[:, e], ey[:, e], p]))
            exp.append(1.0 if o[2 + e] != 0 else 0.0)        # triangle != 0 for checkele = e+1
    got = leaf(2, np.array(rows), 1)[:, 0]
    assert np.array_equal(got, np.array(exp))


def test_snhcsh_bit_exact_or_last_bit(leaf):
    got = leaf(3, G["snhcsh_x"].reshape(-1, 1), 3)
    ref = G["snhcsh_out"]
    # |x| <= .5 is a polynomial (exact); beyond it exp() enters: libm vs device differ in the last bits
    small = np.abs(G["snhcsh_x"]) <= 0.5
    assert np.array_equal(got[small], ref[small])
    assert np.allclose(got[~small], ref[~small], rtol=1e-14, atol=0)


def test_tspack_four_knots(leaf):
    meta, xs, ys = G["sp_meta"], G["sp_x"], G["sp_y"]
    off = ev = 0
    rows, refs = [], []
    for (n, ier, se) in meta:
        if n == 4:
            t = G["sp_eval_t"][ev:ev + 8]
            rows.append(np.concatenate([xs[off:off + 4], ys[off:off + 4], t]))
            refs.append((ier, se, G["sp_yp"][off:off + 4], G["sp_sigma"][off:off + 3], G["sp_eval_out"][ev:ev + 8]))
        off += n
        ev += 8
    got = leaf(4, np.array(rows), 37)
    assert len(rows) > 200
    n_sigerr = n_flip = 0
    for g, (ier, se, yp, sg, evo) in zip(got, refs):
        assert ier == 0
        # YPC1: no transcendental function -> bit-exact, in both forms (the fast path's min / max
        # may only change the sign of a zero)
        assert np.array_equal(g[0:4], yp)
        assert np.array_equal(g[24:28], yp) or np.array_equal(np.abs(g[24:28]), np.abs(yp))
        dev_se = int(g[7])
        if se != 0:
            n_sigerr += 1
        if (dev_se != 0) != (se != 0):
            n_flip += 1            # Newton's last-ulp cycle depends on exp(): DESIGN.md section 2
            continue
        if se != 0:
            continue
        assert np.allclose(g[4:7], sg, rtol=2e-13, atol=0)
        assert np.allclose(g[28:31], sg, rtol=2e-13, atol=0) and g[31] == 0
        assert np.allclose(g[8:16], evo[:, 0], rtol=1e-12, atol=1e-14)       # HVAL
        assert np.allclose(g[16:24], evo[:, 1], rtol=1e-12, atol=1e-14)      # HPVAL
        assert np.allclose(g[32:37], evo[:5, 0], rtol=1e-12, atol=1e-14)     # fast-path HVAL
    assert n_sigerr >= 1 and n_flip <= 2
