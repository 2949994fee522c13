"""Sharding of a batch over ranks (one process per GPU) and the one exchange step of a
multi-pairing: an all-gather of the per-rank Fq12 partial products (72 u64 each).

Independent pairings need no collective: rank r owns the contiguous index range shard_range(n, r, W).
For a multi-pairing product every rank reduces its shard to one Fq12 on its GPU, the partials are
all-gathered (RCCL over xGMI when the process group's backend is "nccl", gloo on CPU in the tests)
and every rank finishes with bls381_finish_multi_pairing.  Field multiplication is not a reduction
RCCL knows, hence gather-then-multiply rather than all-reduce.
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous, balanced [lo, hi) of rank's share of n items."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def all_gather_partials(partial_u64, device=None):
    """partial_u64: numpy (72,) uint64.  Returns (world, 72) uint64 on every rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size()
    t = torch.from_numpy(np.ascontiguousarray(partial_u64, dtype=np.uint64).view(np.int64).copy())
    if device is not None:
        t = t.to(device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return np.stack([o.cpu().numpy().view(np.uint64) for o in outs])
