"""Edge cases of the path: empty and ragged particle sets, nobody released yet, particles
exactly on grid nodes / element edges (the on-node and on-edge branches of gridcell and the
exact-node regime of interp), the smallest level count the spline windows allow."""
import numpy as np
import pytest

import cases
import oracle_lib
import ltrans_amd
from ltrans_amd import roms_synth as rs
from ltrans_amd.driver import LtransRun


def _oracle_kw():
    return dict(lib=oracle_lib.load_oracle(), prefix="lto_")


def _sub(case, idx):
    sub = dict(case)
    sub["cfg"] = case["cfg"].copy()
    sub["cfg"].numpar = len(idx)
    sub["parts"] = {k: (v[idx] if v.ndim == 1 else np.asfortranarray(v[idx])) for k, v in case["parts"].items()}
    return sub


def _run(case, kw, n_external=2, max_internal=4):
    return cases.run_case(case, ltrans_amd.Engine, kw, n_external=n_external, max_internal=max_internal)


def _on_nodes_and_edges(case):
    """move the first particles exactly onto rho nodes, u/v nodes (= rho-element edge
    midpoints) and onto the middle of rho-element diagonals (the two triangles' shared edge)"""
    g = case["g"]
    par = case["parts"]["par"]
    k = 0
    for (i, j) in ((5, 5), (6, 9), (15, 7), (18, 11)):
        par[k, 0], par[k, 1] = g["x_rho"][i, j], g["y_rho"][i, j]; k += 1           # on a rho node
        par[k, 0], par[k, 1] = g["x_u"][i, j], g["y_u"][i, j]; k += 1               # on an edge midpoint
        par[k, 0], par[k, 1] = g["x_v"][i, j], g["y_v"][i, j]; k += 1
        par[k, 0] = 0.5 * (g["x_rho"][i, j] + g["x_rho"][i + 1, j + 1])             # on the diagonal
        par[k, 1] = 0.5 * (g["y_rho"][i, j] + g["y_rho"][i + 1, j + 1]); k += 1
    par[:k, 3:6] = par[:k, 0:3]
    par[:k, 6:9] = par[:k, 0:3]
    geo = case["geo"]
    for key, grid in (("P_r_element", "r"), ("P_u_element", "u"), ("P_v_element", "v")):
        case["parts"][key][:k] = rs.locate_elements(g, geo, par[:k, 0], par[:k, 1], grid)
    return k


# ------------------------------------------------------------------ CPU (oracle)
def test_oracle_nobody_released_and_ragged():
    case = cases.build_case("tiny_turb")
    case["parts"]["par"][:, 10] = 1.0e9          # DOB far in the future
    out = _run(case, _oracle_kw())
    assert out["stats"]["active"] == 0 and out["stats"]["rng_draws"] == 0
    assert np.array_equal(out["par"][:, 0:3], case["parts"]["par"][:, 0:3])
    for n in (1, 63, 257):
        sub = _sub(cases.build_case("tiny_adv"), np.arange(n))
        full = _run(cases.build_case("tiny_adv"), _oracle_kw())
        got = _run(sub, _oracle_kw())
        assert np.array_equal(got["par"], full["par"][:n])       # particles are independent


def test_oracle_on_nodes_and_edges_runs():
    case = cases.build_case("tiny_adv")
    k = _on_nodes_and_edges(case)
    assert (case["parts"]["P_r_element"][:k] > 0).all()
    out = _run(case, _oracle_kw())
    assert np.isfinite(out["par"]).all()


# ------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_empty_ragged_and_unreleased():
    e = ltrans_amd.Engine(ltrans_amd.LtConfig(numpar=0, us=8, ws=9))      # an empty shard
    case = cases.build_case("tiny_adv")
    e.set_grid(case["geo"]); e.set_bounds(case["bounds"])
    e.set_particles(par=np.zeros((0, 13), order="F"), P_r_element=np.zeros(0, np.int32),
                    P_u_element=np.zeros(0, np.int32), P_v_element=np.zeros(0, np.int32))
    run = LtransRun(e, 3600, 120, case["record"])
    run.init_hydro()
    run.run(1, max_internal=2)
    assert e.stats()["active"] == 0
    e.close()
    for n in (1, 63, 257):
        idx = np.arange(n)
        hip = _run(_sub(cases.build_case("tiny_turb"), idx), {})
        orc = _run(_sub(cases.build_case("tiny_turb"), idx), _oracle_kw())
        assert cases.rel_error(hip["par"], orc["par"]) <= 1e-9 and np.array_equal(hip["err_code"], orc["err_code"])
    case = cases.build_case("tiny_turb")
    case["parts"]["par"][:, 10] = 1.0e9
    mt0 = None
    e = cases.make_engine(case, ltrans_amd.Engine, {})
    mt0 = e.get_rng()
    run = LtransRun(e, 3600, 120, case["record"]); run.init_hydro(); run.run(1, max_internal=3)
    assert e.stats()["active"] == 0 and e.stats()["rng_draws"] == 0
    mt1 = e.get_rng()
    assert np.array_equal(mt0[0], mt1[0]) and mt0[1] == mt1[1]           # no deviate consumed
    e.close()


@pytest.mark.gpu
def test_hip_on_nodes_and_edges_parity():
    ch, co = cases.build_case("tiny_adv"), cases.build_case("tiny_adv")
    _on_nodes_and_edges(ch); _on_nodes_and_edges(co)
    hip, orc = _run(ch, {}), _run(co, _oracle_kw())
    assert cases.rel_error(hip["par"], orc["par"]) <= 1e-9
    for k in ("P_r_element", "P_u_element", "P_v_element", "err_code", "hitLand"):
        assert np.array_equal(hip[k], orc[k]), k


@pytest.mark.gpu
def test_hip_minimum_levels():
    """us = 5: the 4-level windows have exactly one position"""
    spec = dict(cases.CASES["tiny_adv"]); spec["grid"] = (24, 16, 5)
    cases.CASES["_tiny_us5"] = spec
    try:
        ch, co = cases.build_case("_tiny_us5"), cases.build_case("_tiny_us5")
        hip, orc = _run(ch, {}), _run(co, _oracle_kw())
    finally:
        del cases.CASES["_tiny_us5"]
    assert cases.rel_error(hip["par"], orc["par"]) <= 1e-9 and np.array_equal(hip["err_code"], orc["err_code"])


def test_get_particles_into_the_arrays_of_an_earlier_call():
    """get_particles(out=...) downloads into the arrays an earlier call returned (the host keeps its arrays between
    print steps) and gives the values a fresh download gives"""
    case = cases.build_case("tiny_adv")
    e = cases.make_engine(case, ltrans_amd.Engine, _oracle_kw())
    run = LtransRun(e, case["cfg"].dt, case["cfg"].idt, case["record"])
    run.init_hydro()
    run.run(1, max_internal=2)
    first = e.get_particles()
    snapshot = {k: v.copy() for k, v in first.items()}
    run.run_external_timestep(2, max_internal=2)
    again = e.get_particles(out=first)
    fresh = e.get_particles()
    for k in fresh:
        assert np.shares_memory(again[k], first[k]), k
        assert again[k].shape == fresh[k].shape and np.array_equal(again[k], fresh[k]), k
    assert not np.array_equal(again["par"], snapshot["par"])
    e.close()
