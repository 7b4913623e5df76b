"""Utterance-shard scatter / gather for multi-GPU restoration (one process per GPU).

Every utterance is restored independently (evaluation_proc/eval.py:119-134 iterates files;
eval-mode BatchNorm has no cross-sample coupling), so the path shards with no data-path
collective: rank r restores clips [lo_r, hi_r).  When the clips originate on one rank (the
reference's handler reads files on the host of a single process), rank 0 deals the shards out
and collects the results with grouped point-to-point sends: over xGMI rank 0 has a direct link
to each of its 7 peers, so a grouped isend/irecv drives all links concurrently (RCCL has no
native scatter).  Works with the `nccl` (= RCCL) backend on GPUs and `gloo` on CPU (tests).
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world):
    """Contiguous, balanced shards: [(lo, hi)] * world."""
    base, rem = divmod(n, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def scatter_clips(full, n, length, device, src=0, dtype=torch.float32):
    """`full` (n, length) on rank `src` (ignored elsewhere) -> this rank's shard (n_r, length)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    bounds = shard_bounds(n, world)
    lo, hi = bounds[rank]
    if rank == src:
        ops = [dist.P2POp(dist.isend, full[a:b].contiguous(), r) for r, (a, b) in enumerate(bounds) if r != src and b > a]
        mine = full[lo:hi].clone()
    else:
        mine = torch.empty((hi - lo, length), device=device, dtype=dtype)
        ops = [dist.P2POp(dist.irecv, mine, src)] if hi > lo else []
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return mine


def gather_clips(shard, n, length, device, dst=0):
    """Inverse of scatter_clips: returns (n, length) on rank `dst`, None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    bounds = shard_bounds(n, world)
    if rank == dst:
        full = torch.empty((n, length), device=device, dtype=shard.dtype)
        lo, hi = bounds[rank]
        full[lo:hi] = shard
        ops = [dist.P2POp(dist.irecv, full[a:b], r) for r, (a, b) in enumerate(bounds) if r != dst and b > a]
    else:
        full = None
        ops = [dist.P2POp(dist.isend, shard.contiguous(), dst)] if shard.shape[0] > 0 else []
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return full


def restore_sharded(engine_fn, full, n, length, device, src=0):
    """scatter -> per-rank restore (engine_fn: (n_r, L) -> (n_r, L)) -> gather on `src`."""
    mine = scatter_clips(full, n, length, device, src)
    out = engine_fn(mine) if mine.shape[0] > 0 else mine
    return gather_clips(out, n, length, device, src)


def selfcheck(device, n=11, length=4096):
    """Round-trip a small tensor through scatter/gather; raises on mismatch."""
    rank = dist.get_rank()
    full = torch.arange(n * length, device=device, dtype=torch.float32).reshape(n, length) if rank == 0 else None
    back = restore_sharded(lambda x: x * 2.0, full, n, length, device)
    if rank == 0 and not torch.equal(back, full * 2.0):
        raise RuntimeError("scatter/gather round trip mismatch")
    return True
