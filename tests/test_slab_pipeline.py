"""The slab pipeline (lt_slab_prefetch / lt_slab_commit = updateHydro's rotation, mask
multiply and copy into t_*(t_f,...), hydrodynamic_module.f90:1052-1410) against the
synchronous lt_set_slab path: same particles after several external steps.  CPU part:
the oracle's reference-shaped mirror of the two entry points."""
import numpy as np
import pytest

import cases
import oracle_lib
import ltrans_amd


def _oracle_kw():
    return dict(lib=oracle_lib.load_oracle(), prefix="lto_")


def _same(a, b):
    assert np.array_equal(a["par"], b["par"])
    for k in ("P_r_element", "P_u_element", "P_v_element", "err_code", "hitBottom", "hitLand"):
        assert np.array_equal(a[k], b[k]), k


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_oracle_pipeline_equals_set_slab(dtype):
    case = cases.build_case("tiny_adv")
    ref = cases.run_case(case, ltrans_amd.Engine, _oracle_kw(), n_external=4, max_internal=4)
    got = cases.run_case(case, ltrans_amd.Engine, _oracle_kw(), n_external=4, max_internal=4, pipeline=dtype)
    _same(got, ref)


def test_oracle_pipeline_window_covers_particles():
    """a window that contains every particle's elements gives the whole-grid result"""
    case = cases.build_case("tiny_adv")
    g = case["g"]
    vi, uj, ui, vj = case["dims"]
    win = [2, vi - 1, 2, uj - 1, 1, ui, 1, uj, 1, vi, 1, vj]   # rho rows/cols at the rim are land
    ref = cases.run_case(case, ltrans_amd.Engine, _oracle_kw(), n_external=4, max_internal=4)
    got = cases.run_case(case, ltrans_amd.Engine, _oracle_kw(), n_external=4, max_internal=4,
                         pipeline=np.float64, window_fn=lambda r: win)
    _same(got, ref)
    assert g["mask_rho"][0, :].max() == 0 and g["mask_rho"][:, 0].max() == 0


def test_oracle_pipeline_misuse():
    case = cases.build_case("tiny_adv")
    e = cases.make_engine(case, ltrans_amd.Engine, _oracle_kw())
    with pytest.raises(ltrans_amd.LtError):
        e.slab_commit(2, 3, 1)                      # nothing prefetched
    e.slab_prefetch(case["rom"](3), case["dims"])
    with pytest.raises(ltrans_amd.LtError):
        e.slab_prefetch(case["rom"](4), case["dims"])   # previous one not committed
    with pytest.raises(ltrans_amd.LtError):
        e.slab_commit(1, 1, 2)                      # not a permutation
    e.slab_commit(2, 3, 1)
    e.close()


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_hip_pipeline_equals_set_slab_and_oracle(dtype):
    case = cases.build_case("tiny_adv")
    sync = cases.run_case(case, ltrans_amd.Engine, {}, n_external=5, max_internal=4)
    pipe = cases.run_case(case, ltrans_amd.Engine, {}, n_external=5, max_internal=4, pipeline=dtype)
    _same(pipe, sync)                              # bit-identical on the device
    orc = cases.run_case(case, ltrans_amd.Engine, _oracle_kw(), n_external=5, max_internal=4, pipeline=dtype)
    assert cases.rel_error(pipe["par"], orc["par"]) <= 1e-9


@pytest.mark.gpu
def test_hip_pipeline_with_turbulence_and_window():
    case = cases.build_case("tiny_turb")
    vi, uj, ui, vj = case["dims"]
    win = [2, vi - 1, 2, uj - 1, 1, ui, 1, uj, 1, vi, 1, vj]
    sync = cases.run_case(case, ltrans_amd.Engine, {}, n_external=4, max_internal=3)
    pipe = cases.run_case(case, ltrans_amd.Engine, {}, n_external=4, max_internal=3,
                          pipeline=np.float32, window_fn=lambda r: win)
    _same(pipe, sync)


@pytest.mark.gpu
def test_hip_pipeline_f64_storage_and_lossless_check():
    case = cases.build_case("tiny_adv", slab_storage=ltrans_amd.abi.LT_SLAB_F64)
    sync = cases.run_case(case, ltrans_amd.Engine, {}, n_external=4, max_internal=3)
    pipe = cases.run_case(case, ltrans_amd.Engine, {}, n_external=4, max_internal=3, pipeline=np.float64)
    _same(pipe, sync)
    # a value that float32 storage cannot hold exactly is reported at the commit
    case32 = cases.build_case("tiny_adv")
    e = cases.make_engine(case32, ltrans_amd.Engine, {})
    rom = {k: (None if v is None else v.copy()) for k, v in case32["rom"](3).items()}
    rom["romU"][3, 3, 2] = 0.1
    e.slab_prefetch(rom, case32["dims"])
    with pytest.raises(ltrans_amd.LtError) as ei:
        e.slab_commit(2, 3, 1)
    assert ei.value.code == ltrans_amd.abi.LT_ERR_SLAB_NOT_F32
    e.close()
