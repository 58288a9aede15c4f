"""world_size-2 run of the multi-GPU host path over gloo on CPU tensors: shard
the particles, advance each shard (the CPU oracle stands in for the per-rank
engine, which needs a GPU), all-gather (x,y,z) after every step and compare with
the unsharded run -- bit-identical, since particles are independent."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nsteps, q):
    sys.path[:0] = [ROOT, HERE]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import cases
    import ltrans_amd
    import oracle_lib
    from ltrans_amd import sharding
    from ltrans_amd.driver import LtransRun
    case = cases.build_case("tiny_adv")
    N = case["cfg"].numpar
    lo, hi = sharding.shard_range(N, world, rank)
    shard = dict(case)
    shard["cfg"] = case["cfg"].copy(numpar=hi - lo)
    shard["parts"] = {k: (v[lo:hi] if v.ndim == 1 else v[lo:hi, :]) for k, v in case["parts"].items()}
    e = cases.make_engine(shard, ltrans_amd.Engine, dict(lib=oracle_lib.load_oracle(), prefix="lto_"))
    run = LtransRun(e, 3600, 120, case["record"])
    run.init_hydro()
    ex = run.external_times(1)
    gathered = None
    for it in range(1, nsteps + 1):
        nact = e.count_active(run.internal_times(ex, it))
        before, total = sharding.rng_window(dist, torch, nact, world, rank)
        e.set_rng_window(before, total)
        e.update_particles(ex, run.internal_times(ex, it), 1, it)
        xyz = e.get_particles(fields=("par",))["par"][:, :3]
        local = torch.from_numpy(np.ascontiguousarray(xyz.T).ravel())      # xyz(n,3) column-major
        gathered = sharding.allgather_positions(dist, torch, local, N, world, rank)
    if rank == 0:
        q.put((gathered.numpy().copy(), before, total))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_run_matches_single_rank():
    sys.path[:0] = [ROOT, HERE]
    import cases
    import ltrans_amd
    import oracle_lib
    nsteps = 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nsteps, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, before, total = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    case = cases.build_case("tiny_adv")
    ref = cases.run_case(case, ltrans_amd.Engine, dict(lib=oracle_lib.load_oracle(), prefix="lto_"),
                         n_external=1, max_internal=nsteps)
    assert np.array_equal(got, ref["par"][:, :3])
    assert (before, total) == (0, case["cfg"].numpar)
