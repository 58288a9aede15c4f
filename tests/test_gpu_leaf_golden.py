"""The DEVICE leaf numerics (lt_math.h: polintd, gridcell, SNHCSH, YPC1 / SIGS / HVAL / HPVAL in
the reference-shaped form and in the forms k_rk's fast path uses) against the vectors the
REFERENCE's own Fortran modules produced (tests/golden/leaf_golden.npz, made with the flang
build of gridcell / interpolation / tension modules): the parity that is pinned to the
reference, checked on the GPU itself and not only through the CPU oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import hip_test_lib
import ltrans_amd

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "leaf_golden.npz"))


@pytest.fixture(scope="module")
def leaf():
    lib = hip_test_lib.load()
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


# ---- the vectors the round-1 suite replayed through the CPU oracle only -------------------------------
@pytest.fixture(scope="module")
def tlib():
    return hip_test_lib.load()


def test_inpoly_reference_vectors_full_and_indexed(tlib):
    """point_in_polygon_module.f90:25-167: every reference vector (all three `onin` modes; points on vertices,
    on edges, on the ray through vertices) through the device's reference-shaped inpoly_full AND through the
    y-slab edge index mbounds / ibounds / psettle use (with its hand-over to the full form)."""
    tlib.lt_test_inpoly.restype = C.c_int
    polys, off = [], 0
    for n in G["ip_npoly"]:
        polys.append(np.ascontiguousarray(G["ip_polys"][off:off + n]))
        off += n
    n_indexed = n_slow = 0
    for pi, P in enumerate(polys):
        sel = G["ip_in"][:, 0] == pi
        q = G["ip_in"][sel]
        exp = G["ip_out"][sel]
        px, py = np.ascontiguousarray(P[:, 0]), np.ascontiguousarray(P[:, 1])
        qx, qy = np.ascontiguousarray(q[:, 1]), np.ascontiguousarray(q[:, 2])
        onin = np.ascontiguousarray(q[:, 3], dtype=np.int32)
        o1, o2, o3 = (np.zeros(q.shape[0], dtype=np.int32) for _ in range(3))
        rc = tlib.lt_test_inpoly(px.ctypes.data_as(C.c_void_p), py.ctypes.data_as(C.c_void_p), C.c_int32(P.shape[0]),
                                 qx.ctypes.data_as(C.c_void_p), qy.ctypes.data_as(C.c_void_p),
                                 onin.ctypes.data_as(C.c_void_p), C.c_int32(q.shape[0]),
                                 o1.ctypes.data_as(C.c_void_p), o2.ctypes.data_as(C.c_void_p), o3.ctypes.data_as(C.c_void_p))
        assert rc == 0
        assert np.array_equal(o1, exp), pi                                  # the full form, every mode
        idx = onin != 0
        assert np.array_equal(o2[idx], exp[idx]), pi                        # the index (on the boundary = inside)
        assert (o2[~idx] == -1).all()
        n_indexed += int(idx.sum())
        n_slow += int(o3.sum())
    assert n_indexed > 1000 and 0 < n_slow < n_indexed       # both the index itself and its hand-over are exercised


def test_linint_reference_vectors(leaf):
    """interpolation_module.f90:25-59 through the strided form VTurb's SigErr fallback uses (linint_n) and, for
    four points, the register form of k_rk_slow (linint4)."""
    rows, off = [], 0
    for k, n in enumerate(G["linint_n"]):
        r = np.zeros(250)
        r[0], r[1] = n, G["linint_t"][k]
        r[2:2 + n] = G["linint_x"][off:off + n]
        r[126:126 + n] = G["linint_y"][off:off + n]
        rows.append(r)
        off += n
    got = leaf(9, np.array(rows), 3)
    assert np.array_equal(got[:, :2], G["linint_out"])
    four = G["linint_n"] == 4
    assert four.sum() > 10 and np.array_equal(got[four, 2], G["linint_out"][four, 0])


def test_tspack_long_profiles_through_the_vturb_forms(tlib):
    """tension_module.f90:231-311,852-1253 on the N = 84 / 124 profiles (N = 4*ws in VTurb): the streaming YPC1 of k_vt_table, its screening + the task-list iteration of
    k_vt_sigs, the [wave][knot][lane] table, and the walk's cached-interval HVAL / HPVAL."""
    tlib.lt_test_spline_n.restype = C.c_int
    meta, xs, ys = G["sp_meta"], G["sp_x"], G["sp_y"]
    off = ev = 0
    rows, refs = [], []
    for (n, ier, se) in meta:
        if n > 4:
            r = np.zeros(257)
            r[0] = n
            r[1:1 + n] = xs[off:off + n]
            r[125:125 + n] = ys[off:off + n]
            r[249:257] = G["sp_eval_t"][ev:ev + 8]
            rows.append(r)
            refs.append((n, ier, se, G["sp_yp"][off:off + n], G["sp_sigma"][off:off + n - 1], G["sp_eval_out"][ev:ev + 8]))
        off += n
        ev += 8
    a = np.ascontiguousarray(np.array(rows))
    out = np.zeros((a.shape[0], 265))
    assert tlib.lt_test_spline_n(a.ctypes.data_as(C.c_void_p), C.c_int32(a.shape[0]), out.ctypes.data_as(C.c_void_p)) == 0
    assert len(rows) == 140
    n_sigerr = n_tension = 0
    for g, (n, ier, se, yp, sg, evo) in zip(out, refs):
        assert ier == 0
        assert np.array_equal(g[0:n], yp)                       # YPC1: no transcendental function -> bit-exact
        assert int(g[248]) == (1 if se != 0 else 0)             # SigErr decided as the reference decides it
        if se != 0:
            n_sigerr += 1
            continue                                            # (the reference then switches to linint)
        assert np.array_equal(g[124:124 + n - 1], sg)           # tension factors (host-libm-exact exp): bit-exact
        n_tension += int((sg != 0).sum())
        assert np.array_equal(g[249:257], evo[:, 0])            # HVAL through vt_seek + hval_prepared
        assert np.array_equal(g[257:265], evo[:, 1])            # HPVAL
    assert n_tension > 50      # (the fixture's SigErr profile is a 4-knot one: test_tspack_four_knots)
