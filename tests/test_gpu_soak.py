"""Parity soak: every case family against the CPU oracle, bit for bit, on particle sets and random streams the other
tests do not use.  LT_SOAK_SEEDS (default 1) is the number of (particle seed, MT19937 seed, shard offset) triples per
family; a deep run (LT_SOAK_SEEDS=30, ~15 min on the box) is a development check, the default keeps the suite short."""
import os

import numpy as np
import pytest

import cases
import ltrans_amd
import oracle_lib
from ltrans_amd import roms_synth as rs

pytestmark = pytest.mark.gpu

NSEEDS = int(os.environ.get("LT_SOAK_SEEDS", "1"))
NMID = int(os.environ.get("LT_SOAK_MID", "0"))      # seeds of the 20000-particle families (the oracle needs a minute each)
FAMILIES = ["tiny_adv", "tiny_sink", "tiny_vt2", "tiny_vt3", "tiny_turb", "tiny_hturb", "tiny_vturb", "tiny_turb_vt2",
            "tiny_turb_vt3", "tiny_beh1", "tiny_beh2", "tiny_beh3", "tiny_beh4", "tiny_beh5", "tiny_beh7", "tiny_salt",
            "tiny_reflect", "tiny_fs_adv", "tiny_fs_turb", "tiny_fs_salt", "tiny_jump", "tiny_open", "tiny_settle"]
INTS = ("P_r_element", "P_u_element", "P_v_element", "err_code", "hitBottom", "hitLand", "dead", "oob", "settle", "endpoly")


def _reseed(case, k):
    """another draw of the particles (same family: same grid, options, fraction near land), another generator seed,
    a number of particles that is not a multiple of the wave size"""
    spec = cases.CASES[case["name"]]
    n = case["cfg"].numpar - 7 * (k % 5) - (k % 3)
    cfg = case["cfg"].copy(numpar=n, seed=9 + 101 * k)
    parts = rs.make_particles(case["g"], case["geo"], case["bounds"], n, seed=777 + 13 * k, near_land=spec["near_land"])
    parts.update(rs.behaviour_state(cfg, n))
    if "age0" in spec:
        lo, hi = spec["age0"]
        parts["par"][:, 11] = np.linspace(lo, hi, n)
        parts["par"][:, 10] = -parts["par"][:, 11] - 1.0
    if k % 4 == 3:      # a few late releases: particles that start drawing mid-run
        parts["par"][5::11, 10] = 120.0 * (1 + (k % 7))
    out = dict(case)
    out["cfg"], out["parts"] = cfg, parts
    return out


@pytest.mark.parametrize("k", range(NSEEDS))
@pytest.mark.parametrize("name", FAMILIES)
def test_soak(name, k):
    case = _reseed(cases.build_case(name), k + 1)
    hip = cases.run_case(case, ltrans_amd.Engine, {}, n_external=2)
    ora = cases.run_case(case, ltrans_amd.Engine, dict(lib=oracle_lib.load_oracle(), prefix="lto_"), n_external=2)
    assert np.array_equal(hip["par"], ora["par"]), (name, k, float(np.abs(hip["par"] - ora["par"]).max()))
    for f in INTS:
        assert np.array_equal(hip[f], ora[f]), (name, k, f)
    assert hip["stats"]["active"] == ora["stats"]["active"]


@pytest.mark.parametrize("k", range(NMID))
@pytest.mark.parametrize("name", ["mid_adv", "mid_turb"])
def test_soak_mid(name, k):
    """the same with 20000 particles (79 workgroups of 256: the kernels' XCD maps with a ragged grid, several generator
    blocks per step, the re-sort) -- opt-in: LT_SOAK_MID=n"""
    case = _reseed(cases.build_case(name), k + 1)
    hip = cases.run_case(case, ltrans_amd.Engine, {}, n_external=2)
    ora = cases.run_case(case, ltrans_amd.Engine, dict(lib=oracle_lib.load_oracle(), prefix="lto_"), n_external=2)
    assert np.array_equal(hip["par"], ora["par"]), (name, k, float(np.abs(hip["par"] - ora["par"]).max()))
    for f in INTS:
        assert np.array_equal(hip[f], ora[f]), (name, k, f)
