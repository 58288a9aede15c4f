"""FreeSlip on the device (k_pre, k_rk's routing of coastal particles to k_rk_slow, wcts_itpi_rho,
k_vt_table) against the oracle: identical trajectories, and the option really changes them."""
import numpy as np
import pytest

import cases
import hip_test_lib
import ltrans_amd
import oracle_lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def okw():
    return dict(lib=oracle_lib.load_oracle(), prefix="lto_")


@pytest.mark.parametrize("name", ["tiny_fs_adv", "tiny_fs_turb", "tiny_fs_salt"])
def test_freeslip_parity(okw, name):
    hip = cases.run_case(cases.build_case(name), ltrans_amd.Engine, {}, n_external=2)
    ora = cases.run_case(cases.build_case(name), ltrans_amd.Engine, okw, n_external=2)
    err = cases.rel_error(hip["par"], ora["par"])
    print(f"{name}: max rel err {err:.3e}")
    assert err <= 1e-6
    for k in ("P_r_element", "P_u_element", "P_v_element", "err_code", "hitBottom", "hitLand", "dead"):
        assert np.array_equal(hip[k], ora[k]), k
    if name == "tiny_fs_salt":
        assert np.allclose(hip["P_Salt"], ora["P_Salt"], rtol=1e-12, atol=0)
        assert np.allclose(hip["P_Temp"], ora["P_Temp"], rtol=1e-12, atol=0)
    off = cases.run_case(cases.build_case(name, FreeSlip=0), ltrans_amd.Engine, {}, n_external=2)
    assert np.count_nonzero(np.any(off["par"][:, :3] != hip["par"][:, :3], axis=1)) > 0


def test_freeslip_needs_masks():
    case = cases.build_case("tiny_fs_adv")
    geo = dict(case["geo"])
    geo["u_mask"] = None
    e = ltrans_amd.Engine(case["cfg"])
    with pytest.raises(ltrans_amd.LtError):
        e.set_grid(geo)


def test_substitution_table_on_the_device():
    """every mask combination of the three variants through the device function, against the
    independent restatement of tests/test_freeslip.py"""
    import ctypes as C
    import test_freeslip as tf
    lib = hip_test_lib.load()
    lib.lt_test_leaf.restype = C.c_int
    v0 = [9.0, 2.0, 7.0, 4.0]
    rows = np.array([[k, m, d] + v0 for k, m, d in tf._all_cases()], dtype=np.float64)
    out = np.empty((rows.shape[0], 4))
    rc = lib.lt_test_leaf(C.c_int32(7), rows.ctypes.data_as(C.c_void_p), C.c_int32(rows.shape[0]), C.c_int32(7),
                          out.ctypes.data_as(C.c_void_p), C.c_int32(4))
    assert rc == 0
    for (k, m, d), got in zip(tf._all_cases(), out):
        mm = [(m >> b) & 1 for b in range(4)]
        dd = [(d >> b) & 1 for b in range(4)]
        assert list(got) == tf._py_sub(k, mm, dd, v0), (k, mm, dd)
