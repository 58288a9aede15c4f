"""Multi-process runs of the multi-GPU host path over gloo on CPU tensors: shard the particles, advance each
shard (the CPU oracle stands in for the per-rank engine, which needs a GPU), exchange the window of the serial
random stream before every step (LTRANS.f90:790-816, 1087-1098: one generator, consumed in particle order by the
particles that are tracked), all-gather (x,y,z) after every step, and compare with the unsharded run: bit-identical
positions and the same final generator state on every rank.

The case has HTurb + VTurb on and a fifth of the first half of the particles released only at the third step, so
that the ranks hold unequal numbers of drawing particles, the numbers change during the run, and a wrong window
changes the result."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
NSTEPS = 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    import cases
    case = cases.build_case("tiny_turb")
    par = case["parts"]["par"]
    n = par.shape[0]
    late = (np.arange(n) % 5 == 0) & (np.arange(n) < n // 2)
    par[late, 10] = 250.0            # par(:,pDOB): first tracked at the third step (ix(3) = 360 > 250)
    return case, int(late.sum())


def _next_draws(oracle, mt, mti, n=8):
    import ctypes as C
    import oracle_lib
    st = oracle_lib.MT()
    for i in range(624):
        st.mt[i] = int(mt[i])
    st.mti = int(mti)
    return [int(oracle.lto_genrand_int32(C.byref(st))) for _ in range(n)]


def _worker(rank, world, port, sizes, q):
    sys.path[:0] = [ROOT, HERE]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cases
    import ltrans_amd
    import oracle_lib
    from ltrans_amd import sharding
    from ltrans_amd.driver import LtransRun
    case, _ = _case()
    N = case["cfg"].numpar
    lo = int(np.sum(sizes[:rank]))
    hi = lo + sizes[rank]
    shard = dict(case)
    shard["cfg"] = case["cfg"].copy(numpar=hi - lo)
    shard["parts"] = {k: (v[lo:hi] if v.ndim == 1 else v[lo:hi, :]) for k, v in case["parts"].items()}
    oracle = oracle_lib.load_oracle()
    e = cases.make_engine(shard, ltrans_amd.Engine, dict(lib=oracle, prefix="lto_"))
    run = LtransRun(e, 3600, 120, case["record"])
    run.init_hydro()
    ex = run.external_times(1)
    gathered, windows = None, []
    for it in range(1, NSTEPS + 1):
        nact = e.count_active(run.internal_times(ex, it))
        before, total = sharding.rng_window(dist, torch, nact, world, rank)
        windows.append((nact, before, total))
        e.set_rng_window(before, total)
        e.update_particles(ex, run.internal_times(ex, it), 1, it)
        xyz = e.get_particles(fields=("par",))["par"][:, :3]
        local = torch.from_numpy(np.ascontiguousarray(xyz.T).ravel())      # xyz(n,3) column-major
        gathered = sharding.allgather_positions(dist, torch, local, N, world, rank, sizes=sizes)
    q.put((rank, gathered.numpy().copy() if rank == 0 else None, windows, _next_draws(oracle, *e.get_rng())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("sizes", [(256, 256), (300, 0, 212)], ids=["two_ranks", "three_ranks_one_empty"])
def test_sharded_turbulence_run_matches_single_rank(sizes):
    sys.path[:0] = [ROOT, HERE]
    import cases
    import ltrans_amd
    import oracle_lib
    from ltrans_amd.driver import LtransRun
    world = len(sizes)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, list(sizes), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, got, windows, nxt = q.get(timeout=300)
        res[r] = (got, windows, nxt)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the unsharded run
    case, nlate = _case()
    N = case["cfg"].numpar
    assert sum(sizes) == N
    oracle = oracle_lib.load_oracle()
    e = cases.make_engine(case, ltrans_amd.Engine, dict(lib=oracle, prefix="lto_"))
    run = LtransRun(e, 3600, 120, case["record"])
    run.init_hydro()
    run.run(1, max_internal=NSTEPS)
    ref = e.get_particles(fields=("par",))["par"][:, :3]
    ref_next = _next_draws(oracle, *e.get_rng())
    e.close()
    assert np.array_equal(res[0][0], ref)
    # every rank ends at the unsharded run's position of the stream
    for r in range(world):
        assert res[r][2] == ref_next, r
    # the windows that were exchanged: unequal shards of the stream, changing when the late particles are released
    for it in range(NSTEPS):
        total = N - nlate if it < 2 else N
        acts = [res[r][1][it][0] for r in range(world)]
        assert sum(acts) == total
        for r in range(world):
            assert res[r][1][it][1:] == (sum(acts[:r]), total)
    assert [res[r][1][0][0] for r in range(world)] != [res[r][1][2][0] for r in range(world)]
    if world == 2:
        assert res[0][1][0][0] != res[1][1][0][0]      # unequal numbers of drawing particles on the two ranks
    # and a window that is wrong does change the result (the test would not notice a window that is ignored)
    lo = sizes[0]
    shard = dict(case)
    case2, _ = _case()
    shard["cfg"] = case2["cfg"].copy(numpar=N - lo)
    shard["parts"] = {k: (v[lo:] if v.ndim == 1 else v[lo:, :]) for k, v in case2["parts"].items()}
    e = cases.make_engine(shard, ltrans_amd.Engine, dict(lib=oracle, prefix="lto_"))
    run = LtransRun(e, 3600, 120, case2["record"])
    run.init_hydro()
    ex = run.external_times(1)
    e.set_rng_window(0, N - nlate)           # as if no particle of a lower rank drew
    e.update_particles(ex, run.internal_times(ex, 1), 1, 1)
    wrong = e.get_particles(fields=("par",))["par"][:, :3]
    e.close()
    case3, _ = _case()
    e = cases.make_engine(case3, ltrans_amd.Engine, dict(lib=oracle, prefix="lto_"))
    run = LtransRun(e, 3600, 120, case3["record"])
    run.init_hydro()
    run.run(1, max_internal=1)
    one = e.get_particles(fields=("par",))["par"][lo:, :3]
    e.close()
    assert not np.array_equal(wrong, one)
