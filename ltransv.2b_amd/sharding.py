"""Particle sharding across the GPUs of one node.

Particles are independent within a step (LTRANS.f90:778-1400 touches only index
n), so rank g of G holds the contiguous index block [g*N/G, (g+1)*N/G); ROMS
slabs, grid, boundaries and habitat are replicated.  The data path needs no
collective; what the ranks exchange is
  * the new (x,y,z) of every shard at the end of a step (all-gather, RCCL over
    xGMI when the tensors live on GPUs) -- the reference needs globally
    assembled positions only at print steps (LTRANS.f90:604-612);
  * when turbulence is on, one int64 per rank before the step: the number of
    particles that will draw from the single serial Mersenne-Twister stream,
    from which each rank derives its window of the stream.
All functions take a ``torch.distributed``-like module so that the same code
runs over gloo on CPU tensors in the tests.
"""
import numpy as np


def shard_range(numpar, world, rank):
    """[lo, hi) of the global particle indices (0-based) held by ``rank``."""
    base, rem = divmod(int(numpar), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_sizes(numpar, world):
    return [shard_range(numpar, world, r)[1] - shard_range(numpar, world, r)[0]
            for r in range(world)]


def allgather_positions(dist, torch, local_xyz, numpar_global, world, rank, sizes=None):
    """local_xyz: tensor (3*n_local,) = xyz(n_local,3) column-major of this shard.
    Returns a (numpar_global, 3) tensor in global particle order on every rank.
    sizes: the shards' particle counts when they are not shard_sizes() (ragged or empty shards)."""
    sizes = shard_sizes(numpar_global, world) if sizes is None else [int(m) for m in sizes]
    assert len(sizes) == world and sum(sizes) == numpar_global
    nmax = max(sizes)
    send = local_xyz.new_zeros(3 * nmax)
    n = sizes[rank]
    send.view(3, nmax)[:, :n] = local_xyz.view(3, n)
    recv = local_xyz.new_empty(world * 3 * nmax)
    if world > 1:
        dist.all_gather_into_tensor(recv, send)
    else:
        recv.copy_(send)
    out = local_xyz.new_empty(numpar_global, 3)
    lo = 0
    for r, m in enumerate(sizes):
        out[lo:lo + m] = recv[r * 3 * nmax:(r + 1) * 3 * nmax].view(3, nmax)[:, :m].t()
        lo += m
    return out


def rng_window(dist, torch, n_active_local, world, rank, device="cpu"):
    """(active particles held by lower ranks, total active) for lt_set_rng_window."""
    mine = torch.tensor([int(n_active_local)], dtype=torch.int64, device=device)
    if world > 1:
        allc = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine)
        counts = torch.cat(allc).tolist()          # one device-to-host read for all ranks
    else:
        counts = [int(n_active_local)]
    return int(np.sum(counts[:rank])), int(np.sum(counts))
