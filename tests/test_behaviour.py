"""Biological behaviours 1-5 and 7 and SaltTempOn (behavior_module.f90:181-551,
LTRANS.f90:1062-1076).

CPU part: properties of the oracle's restatement (draw accounting, lto_count_draws leaves
the state alone, sharded windows in draws).  GPU part: the HIP path against the oracle --
same positions (the deviates, the salinity cue and, since the kernels evaluate libm's functions
exactly as the host does, every bit of them), identical integer state and identical numbers of
deviates drawn."""
import numpy as np
import pytest

import cases
import oracle_lib
import ltrans_amd
from ltrans_amd.driver import LtransRun

BEH_CASES = ["tiny_beh1", "tiny_beh2", "tiny_beh3", "tiny_beh4", "tiny_beh5", "tiny_beh4_old",
             "tiny_beh5_old", "tiny_beh7", "tiny_salt"]
INT_STATE = ("P_r_element", "P_u_element", "P_v_element", "P_behave", "dead", "oob", "bottom",
             "hitBottom", "hitLand", "err_code")


def _oracle_kw():
    return dict(lib=oracle_lib.load_oracle(), prefix="lto_")


def _run(case, kw, n_external=3, max_internal=5, draws=None):
    e = cases.make_engine(case, ltrans_amd.Engine, kw)
    run = LtransRun(e, case["cfg"].dt, case["cfg"].idt, case["record"])
    run.init_hydro()

    def on_step(p, it, ex, ix):
        if draws is not None:
            draws.append(e.stats()["rng_draws"])

    run.run(n_external, max_internal=max_internal, on_step=on_step)
    out = e.get_particles()
    e.close()
    return out


# ------------------------------------------------------------------ CPU
def test_oracle_draw_accounting():
    # types 1 and 2: always two deviates per tracked particle; with HTurb four more
    d1, d2 = [], []
    _run(cases.build_case("tiny_beh1"), _oracle_kw(), 2, 3, d1)
    _run(cases.build_case("tiny_beh2"), _oracle_kw(), 2, 3, d2)
    assert set(d1) == {2 * 512} and set(d2) == {6 * 512}
    # type 5 with young larvae: 2 deviates, or 3 when the salinity cue fires before 3.5 days
    d5 = []
    out = _run(cases.build_case("tiny_beh5"), _oracle_kw(), 3, 5, d5)
    assert min(d5) >= 2 * 512 and max(d5) > 2 * 512 and max(d5) <= 3 * 512
    assert (out["timer"] == 0).all()
    # old C. ariakensis: the cue starts the 2-hour timer, pediveligers change behaviour
    out = _run(cases.build_case("tiny_beh5_old"), _oracle_kw(), 3, 5)
    assert (out["timer"] > 0).any() and (out["P_behave"] == 2).any() and (out["P_behave"] == 5).any()
    out = _run(cases.build_case("tiny_beh4_old"), _oracle_kw(), 3, 5)
    assert (out["P_behave"] == 2).any() and (out["P_behave"] == 4).any()
    # tidal stream transport draws nothing and lifts particles off the bottom
    d7 = []
    out = _run(cases.build_case("tiny_beh7"), _oracle_kw(), 3, 5, d7)
    assert set(d7) == {0} and 0 < out["bottom"].sum() < 512


def test_oracle_count_draws_is_a_dry_run_and_windows_in_draws():
    case = cases.build_case("tiny_beh5")
    ref = _run(case, _oracle_kw(), 2, 4)
    # the same run, asking for the draws before every step
    e = cases.make_engine(case, ltrans_amd.Engine, _oracle_kw())
    run = LtransRun(e, case["cfg"].dt, case["cfg"].idt, case["record"])
    run.init_hydro()
    for p in (1, 2):
        if p > 2:
            run.update_hydro()
        ex = run.external_times(p)
        for it in range(1, 5):
            ix = run.internal_times(ex, it)
            n = e.count_draws(ex, ix, p, it)
            e.update_particles(ex, ix, p, it)
            assert e.stats()["rng_draws"] == n
    got = e.get_particles()
    e.close()
    assert np.array_equal(got["par"], ref["par"])
    # two shards with windows in draws reproduce the single run
    n = case["cfg"].numpar
    half = n // 2
    shards = []
    for lo, hi in ((0, half), (half, n)):
        sub = dict(case)
        sub["cfg"] = case["cfg"].copy()
        sub["cfg"].numpar = hi - lo
        sub["parts"] = {k: (v[lo:hi] if v.ndim == 1 else np.asfortranarray(v[lo:hi])) for k, v in case["parts"].items()}
        es = cases.make_engine(sub, ltrans_amd.Engine, _oracle_kw())
        rs_ = LtransRun(es, case["cfg"].dt, case["cfg"].idt, case["record"])
        rs_.init_hydro()
        shards.append((es, rs_))
    for p in (1, 2):
        ex = shards[0][1].external_times(p)
        for it in range(1, 5):
            ix = shards[0][1].internal_times(ex, it)
            counts = [es.count_draws(ex, ix, p, it) for es, _ in shards]
            for r, (es, _) in enumerate(shards):
                es.set_rng_window_draws(sum(counts[:r]), sum(counts))
                es.update_particles(ex, ix, p, it)
    par = np.vstack([es.get_particles()["par"] for es, _ in shards])
    for es, _ in shards:
        es.close()
    assert np.array_equal(par, ref["par"])


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", BEH_CASES)
def test_hip_behaviour_parity(name):
    case = cases.build_case(name)
    do, dh = [], []
    orc = _run(case, _oracle_kw(), 3, 5, do)
    hip = _run(case, {}, 3, 5, dh)
    assert dh == do                                   # identical numbers of deviates, every step
    for k in INT_STATE:
        assert np.array_equal(hip[k], orc[k]), k
    assert cases.rel_error(hip["par"], orc["par"]) <= 1e-9
    for k in ("timer", "P_Sprev", "P_zprev", "P_Salt", "P_Temp"):
        assert np.allclose(hip[k], orc[k], rtol=1e-9, atol=1e-12), k
    assert np.allclose(hip["P_swim"], orc["P_swim"], rtol=1e-12, atol=0)


@pytest.mark.gpu
def test_hip_behaviour_sharded_draw_windows():
    case = cases.build_case("tiny_beh5")
    ref = _run(case, {}, 2, 4)
    n = case["cfg"].numpar
    half = n // 2
    shards = []
    for lo, hi in ((0, half), (half, n)):
        sub = dict(case)
        sub["cfg"] = case["cfg"].copy()
        sub["cfg"].numpar = hi - lo
        sub["parts"] = {k: (v[lo:hi] if v.ndim == 1 else np.asfortranarray(v[lo:hi])) for k, v in case["parts"].items()}
        es = cases.make_engine(sub, ltrans_amd.Engine, {})
        r = LtransRun(es, case["cfg"].dt, case["cfg"].idt, case["record"])
        r.init_hydro()
        shards.append((es, r))
    for p in (1, 2):
        ex = shards[0][1].external_times(p)
        for it in range(1, 5):
            ix = shards[0][1].internal_times(ex, it)
            counts = [es.count_draws(ex, ix, p, it) for es, _ in shards]
            for r, (es, _) in enumerate(shards):
                es.set_rng_window_draws(sum(counts[:r]), sum(counts))
                es.update_particles(ex, ix, p, it)
    par = np.vstack([es.get_particles()["par"] for es, _ in shards])
    for es, _ in shards:
        es.close()
    assert np.array_equal(par, ref["par"])


@pytest.mark.gpu
def test_particle_window_refused_with_behaviour_draws():
    case = cases.build_case("tiny_beh1")
    e = cases.make_engine(case, ltrans_amd.Engine, {})
    with pytest.raises(ltrans_amd.LtError):
        e.set_rng_window(0, 10)
    e.close()
