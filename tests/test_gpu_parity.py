"""GPU parity tests proper: the HIP engine, called through the C ABI, against the
CPU oracle on identical seeded inputs.  Tolerance: BASELINE.json north_star
"<= 1e-6 relative on particle (x,y,z)"; integer state (element ids, error codes,
collision counters) must be identical."""
import numpy as np
import pytest

import cases
import ltrans_amd
import oracle_lib

pytestmark = pytest.mark.gpu

RTOL = 1e-6


def _both(case, **kw):
    hip = cases.run_case(case, ltrans_amd.Engine, {}, **kw)
    ora = cases.run_case(case, ltrans_amd.Engine,
                         dict(lib=oracle_lib.load_oracle(), prefix="lto_"), **kw)
    return hip, ora


def _check(hip, ora):
    err = cases.rel_error(hip["par"], ora["par"])
    assert err <= RTOL, f"max rel err {err:.3e}"
    for k in ("P_r_element", "P_u_element", "P_v_element", "err_code", "hitBottom", "hitLand",
              "dead", "oob", "settle", "endpoly"):
        assert np.array_equal(hip[k], ora[k]), k
    # previous / new positions and ages too
    np.testing.assert_allclose(hip["par"][:, 3:9], ora["par"][:, 3:9], rtol=RTOL, atol=1e-6)
    assert np.array_equal(hip["par"][:, 9:], ora["par"][:, 9:])
    assert hip["stats"]["active"] == ora["stats"]["active"]
    return err


@pytest.mark.parametrize("name", ["tiny_adv", "tiny_sink", "small_adv", "tiny_vt2", "tiny_vt3"])
def test_parity_two_external_steps(name):
    case = cases.build_case(name)
    hip, ora = _both(case, n_external=2)
    err = _check(hip, ora)
    print(f"{name}: max rel err {err:.3e}; land hits {int(ora['hitLand'].sum())}")


@pytest.mark.parametrize("errflag", [1, 2, 3])
def test_parity_reflection(errflag):
    """drift into coast and island: intersect_reflect, up to 3 reflections, error paths."""
    case = cases.build_case("tiny_reflect", ErrorFlag=errflag)
    hip, ora = _both(case, n_external=2)
    _check(hip, ora)
    assert int(ora["hitLand"].sum()) > 50, "case must exercise reflections"


@pytest.mark.parametrize("errflag", [1, 2, 3])
def test_parity_jumped_over_an_element(errflag):
    """steps longer than a cell: the closing setEle fails (27-29) AFTER par(n,pn*) took the new position
    (LTRANS.f90:1312-1317), so under ErrorFlag 2 / 3 the dead / out-of-bounds particle is moved there."""
    case = cases.build_case("tiny_jump", ErrorFlag=errflag, mortality=1 if errflag == 2 else 0, deadage=1e9,
                            OpenOceanBoundary=1 if errflag == 3 else 0)
    hip, ora = _both(case, n_external=1, max_internal=6)
    _check(hip, ora)
    if errflag == 2:
        assert int(ora["dead"].sum()) > 100
    if errflag == 3:
        assert int(ora["oob"].sum()) > 100


def test_parity_open_ocean_boundary():
    case = cases.build_case("tiny_open")
    hip, ora = _both(case, n_external=3)
    _check(hip, ora)
    assert int(ora["oob"].sum()) > 10, "case must let particles leave through the open boundary"


def test_parity_settlement_and_mortality():
    case = cases.build_case("tiny_settle")
    hip, ora = _both(case, n_external=3)
    _check(hip, ora)
    assert int(ora["settle"].sum()) > 5 and int(ora["dead"].sum()) > 5
    assert np.array_equal(hip["par"][:, 12], ora["par"][:, 12])      # lifespan


def test_parity_slab_rotation():
    """p = 1 special case, p = 2, and two updateHydro rotations (p = 3, 4)."""
    case = cases.build_case("tiny_adv")
    hip, ora = _both(case, n_external=4, max_internal=8)
    _check(hip, ora)


def test_parity_f64_slabs():
    case = cases.build_case("tiny_adv", slab_storage=ltrans_amd.abi.LT_SLAB_F64)
    hip, ora = _both(case, n_external=2, max_internal=10)
    _check(hip, ora)


def test_parity_mid():
    case = cases.build_case("mid_adv")
    hip, ora = _both(case, n_external=1)
    _check(hip, ora)


def test_index_equals_brute_force(monkeypatch):
    """boundary indexes vs the reference's full scans (LT_BRUTE_BOUNDS=1): bit-identical."""
    case = cases.build_case("tiny_adv")
    a = cases.run_case(case, ltrans_amd.Engine, {}, n_external=2)
    monkeypatch.setenv("LT_BRUTE_BOUNDS", "1")
    b = cases.run_case(case, ltrans_amd.Engine, {}, n_external=2)
    assert np.array_equal(a["par"], b["par"])
    assert np.array_equal(a["hitLand"], b["hitLand"])


def test_abi_allgather_single_rank(tmp_path):
    """the RCCL all-gather behind the ABI (lt_comm_init_file / lt_allgather_positions), one rank: the gathered
    xyz is the shard's own (x,y,z) in the reference's particle order (the N-rank form needs N GPUs)."""
    case = cases.build_case("tiny_adv")
    e = cases.make_engine(case, ltrans_amd.Engine, {})
    from ltrans_amd.driver import LtransRun
    run = LtransRun(e, case["cfg"].dt, case["cfg"].idt, case["record"])
    run.init_hydro()
    run.run(1, max_internal=3)
    idfile = tmp_path / "lt_comm_id"
    e.comm_init_file(idfile, 1, 0)
    assert not idfile.exists()          # rank 0 removes the id file once the communicator exists
    xyz = e.allgather_positions([case["cfg"].numpar])
    assert np.array_equal(xyz, e.get_particles(fields=("par",))["par"][:, :3])
    # the asynchronous form: the next step is enqueued behind the call and runs beside the collective
    import torch
    dst = torch.empty(3 * case["cfg"].numpar, dtype=torch.float64, device="cuda")
    before = e.get_particles(fields=("par",))["par"][:, :3].copy()
    e.allgather_positions_device([case["cfg"].numpar], dst.data_ptr())
    run.run_external_timestep(2, max_internal=1)
    e.allgather_wait()
    torch.cuda.synchronize()
    assert np.array_equal(dst.cpu().numpy().reshape(3, -1).T, before)
    e.comm_destroy()
    e.close()


def test_comm_id_file_of_an_earlier_run_is_ignored(tmp_path):
    """a rank > 0 does not take the id file a crashed run left behind (older than its own start - timeout)"""
    import os
    import time
    case = cases.build_case("tiny_adv")
    e = cases.make_engine(case, ltrans_amd.Engine, {})
    stale = tmp_path / "lt_comm_id"
    stale.write_bytes(b"\0" * 128)
    old = time.time() - 3600
    os.utime(stale, (old, old))
    with pytest.raises(RuntimeError, match="did not appear"):
        e.comm_init_file(stale, 2, 1, timeout_s=1)
    # a long-lived process: a file of an attempt ten seconds ago (after this library was loaded, for all it knows)
    # is as stale as one of an hour ago -- the age counts from the call
    recent = time.time() - 10
    os.utime(stale, (recent, recent))
    with pytest.raises(RuntimeError, match="did not appear"):
        e.comm_init_file(stale, 2, 1, timeout_s=1)
    e.close()


def test_allgather_layout_three_ragged_shards():
    """The N-rank form of lt_allgather_positions on one GPU (test build: the collective is replaced by what it
    delivers): padded blocks [rank][3][nmax] of three shards of unequal size -- one of them empty -- assembled in
    global particle order, for this engine as the first, the middle and the last rank."""
    import hip_test_lib
    case = cases.build_case("tiny_adv")
    n = case["cfg"].numpar
    rng = np.random.default_rng(5)
    for rank, counts in ((0, [n, 0, 3 * n + 7]), (1, [5, n, 2 * n + 1]), (2, [n + 9, 1, n]), (1, [0, n, 0])):
        e = cases.make_engine(case, hip_test_lib.HipTestEngine, {})
        own = e.get_particles(fields=("par",))["par"][:, :3]
        n_others = sum(c for r, c in enumerate(counts) if r != rank)
        others = np.asfortranarray(rng.standard_normal((n_others, 3)))
        got = e.allgather_layout(3, rank, counts, others)
        parts, o = [], 0
        for r, c in enumerate(counts):
            if r == rank:
                parts.append(own)
            else:
                parts.append(others[o:o + c])
                o += c
        assert np.array_equal(got, np.concatenate(parts, axis=0)), (rank, counts)
        e.close()


def test_downloads_through_the_staging_buffer(oracle):
    """lt_get_particles through the library's pinned staging buffer: a download of one array (small buffer), then of
    everything (the buffer grows), then into the arrays of the earlier call, with few and with many host threads --
    always the oracle's values"""
    import os
    case = cases.build_case("tiny_sink")
    outs = {}
    for which in ("hip", "oracle"):
        e = cases.make_engine(case, ltrans_amd.Engine, {} if which == "hip" else dict(lib=oracle, prefix="lto_"))
        from ltrans_amd.driver import LtransRun
        run = LtransRun(e, case["cfg"].dt, case["cfg"].idt, case["record"])
        run.init_hydro()
        run.run(1, max_internal=3)
        first = e.get_particles(fields=("par",))
        full = e.get_particles()
        assert np.array_equal(first["par"], full["par"])
        os.environ["LT_GET_THREADS"] = "3"
        try:
            again = e.get_particles(out=full)
        finally:
            del os.environ["LT_GET_THREADS"]
        for k in full:
            assert np.shares_memory(again[k], full[k])
        outs[which] = {k: v.copy() for k, v in again.items()}
        e.close()
    for k in outs["oracle"]:
        assert np.array_equal(outs["hip"][k], outs["oracle"][k]), k
