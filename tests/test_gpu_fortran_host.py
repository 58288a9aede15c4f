"""The Fortran host harness (ISO_C_BINDING over the C ABI) must reproduce the
Python-driven engine bit for bit: same library, same arrays, same loops."""
import os
import subprocess

import numpy as np
import pytest

import cases
import ltrans_amd
from ltrans_amd import casefile

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "ltransv.2b_amd", "fortran", "ltrans_hip_driver")


@pytest.mark.parametrize("name,pipeline", [("tiny_adv", 0), ("tiny_sink", 0), ("tiny_adv", 1), ("tiny_turb", 1)])
def test_fortran_driver_matches_python_driver(tmp_path, name, pipeline):
    if not os.path.exists(DRIVER):
        pytest.skip("Fortran harness not built (flang absent at build time)")
    case = cases.build_case(name)
    n_ext, n_int = 3, 7
    casefile.write_ltrans_data(tmp_path / "LTRANS.data", case["cfg"], days=1.0)
    casefile.write_case(tmp_path / "case.bin", case["g"], case["geo"], case["bounds"], case["parts"],
                        [case["record"](r) for r in range(n_ext + 1)], case["cfg"].numpar)
    r = subprocess.run([DRIVER, str(tmp_path / "LTRANS.data"), str(tmp_path / "case.bin"),
                        str(tmp_path / "out.bin"), str(n_ext), str(n_int), str(pipeline)], capture_output=True, text=True,
                       cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    f = casefile.read_result(tmp_path / "out.bin")
    assert "setEle_all on the device: 0 particles located differently" in r.stdout
    p = cases.run_case(case, ltrans_amd.Engine, {}, n_external=n_ext, max_internal=n_int)
    assert f["steps"] == n_ext * n_int
    assert np.array_equal(f["par"], p["par"])
    for k in ("P_r_element", "P_u_element", "P_v_element", "err_code", "hitBottom", "hitLand"):
        assert np.array_equal(f[k], p[k]), k
    # the print step through lt_get_output, from Fortran
    e = cases.make_engine(cases.build_case(name), ltrans_amd.Engine, {})
    e.set_particles(**{k: p[k] for k in ("par", "P_r_element", "P_u_element", "P_v_element", "dead", "oob", "settle")})
    o = e.get_output()
    e.close()
    for k in ("lon", "lat", "z", "age", "status"):
        assert np.array_equal(f[k], o[k]), k


def test_fortran_driver_rank_mode(tmp_path):
    """the harness as one process per GPU (LT_NRANKS / LT_RANK / LT_COMM_FILE), here with one rank: the shard is the
    whole case, the static stream window is the whole stream, and the array the all-gather returns is par(:,1:3)."""
    if not os.path.exists(DRIVER):
        pytest.skip("Fortran harness not built (flang absent at build time)")
    case = cases.build_case("tiny_turb")
    n_ext, n_int = 2, 5
    casefile.write_ltrans_data(tmp_path / "LTRANS.data", case["cfg"], days=1.0)
    casefile.write_case(tmp_path / "case.bin", case["g"], case["geo"], case["bounds"], case["parts"],
                        [case["record"](r) for r in range(n_ext + 1)], case["cfg"].numpar)
    env = dict(os.environ, LT_NRANKS="1", LT_RANK="0", LT_COMM_FILE=str(tmp_path / "comm_id"), LT_DEVICE="0")
    r = subprocess.run([DRIVER, str(tmp_path / "LTRANS.data"), str(tmp_path / "case.bin"),
                        str(tmp_path / "out.bin"), str(n_ext), str(n_int), "0"], capture_output=True, text=True,
                       cwd=tmp_path, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    f = casefile.read_result(tmp_path / "out.bin.rank0")
    p = cases.run_case(case, ltrans_amd.Engine, {}, n_external=n_ext, max_internal=n_int)
    assert np.array_equal(f["par"], p["par"])
    assert np.array_equal(f["xyz_all"], p["par"][:, :3])
    assert not (tmp_path / "comm_id").exists()


def test_fortran_driver_rank_mode_needs_a_communicator_for_several_ranks(tmp_path):
    """LT_NRANKS > 1 without LT_COMM_FILE: the harness stops before it creates anything (the shards could neither
    exchange their draw counts -- lt_exchange_rng_window, tests/test_gpu_exchange.py -- nor gather positions)"""
    if not os.path.exists(DRIVER):
        pytest.skip("Fortran harness not built (flang absent at build time)")
    case = cases.build_case("tiny_turb", mortality=1, deadage=1.0e9)
    casefile.write_ltrans_data(tmp_path / "LTRANS.data", case["cfg"], days=1.0)
    casefile.write_case(tmp_path / "case.bin", case["g"], case["geo"], case["bounds"], case["parts"],
                        [case["record"](r) for r in range(2)], case["cfg"].numpar)
    env = {k: v for k, v in os.environ.items() if k != "LT_COMM_FILE"}
    env.update(LT_NRANKS="2", LT_RANK="1", LT_DEVICE="0")
    r = subprocess.run([DRIVER, str(tmp_path / "LTRANS.data"), str(tmp_path / "case.bin"),
                        str(tmp_path / "out.bin"), "1", "3", "0"], capture_output=True, text=True, cwd=tmp_path, env=env,
                       timeout=120)
    assert r.returncode != 0
    assert "needs LT_COMM_FILE" in r.stdout + r.stderr
    assert not (tmp_path / "out.bin.rank1").exists() and not (tmp_path / "out.bin").exists()


def test_fortran_driver_writes_errorlog_and_timing(tmp_path):
    """the harness's ErrorLog.txt (one line per particle error in the reference's formats, LTRANS.f90:761-775,
    858-877) and, with WriteModelTiming, Timing.csv (header :471-476, one row per print interval :1669-1689)."""
    if not os.path.exists(DRIVER):
        pytest.skip("Fortran harness not built (flang absent at build time)")
    case = cases.build_case("tiny_jump", ErrorFlag=1)
    n_ext, n_int = 1, 6
    casefile.write_ltrans_data(tmp_path / "LTRANS.data", case["cfg"], days=1.0, iprint=2 * case["cfg"].idt, model_timing=True)
    casefile.write_case(tmp_path / "case.bin", case["g"], case["geo"], case["bounds"], case["parts"],
                        [case["record"](r) for r in range(n_ext + 2)], case["cfg"].numpar)
    r = subprocess.run([DRIVER, str(tmp_path / "LTRANS.data"), str(tmp_path / "case.bin"), str(tmp_path / "out.bin"),
                        str(n_ext), str(n_int), "0"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    total = int(r.stdout.strip().splitlines()[-1].rsplit(":", 1)[1])
    trace = []
    cases.run_case(case, ltrans_amd.Engine, {}, n_external=n_ext, max_internal=n_int, trace=trace)
    codes = np.concatenate([t["err_code"] for t in trace])
    assert total == int((codes != 0).sum()) and total > 100
    lines = (tmp_path / "ErrorLog.txt").read_text().splitlines()
    assert len(lines) == total
    want = {28: "jumped over u element", 29: "jumped over v element", 27: "jumped over rho element",
            25: "outside main bounds after intersect_reflect", 26: "inside island bounds after intersect_reflect",
            24: "out after 3rd reflection"}
    for code, text in want.items():
        assert sum(text in ln for ln in lines) == int((codes == code).sum()), code
    first = np.flatnonzero(trace[0]["err_code"])[0]
    assert lines[0].startswith("Particle %10d" % (first + 1)) and lines[0].endswith("%10d seconds" % 120)
    rows = (tmp_path / "Timing.csv").read_text().splitlines()
    assert rows[0].strip().startswith("Daytime,Elapsed,Hydro,Hydro %,Set Up") and len(rows) == 1 + n_int // 2
    vals = [float(v) for v in rows[1].split(",")]
    assert len(vals) == 16 and vals[1] > 0 and vals[6] > 0 and vals[14] > 0      # elapsed, advection, update
