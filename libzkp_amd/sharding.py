"""Multi-GPU form of process_batch for the range path: one process per GPU, independent ops sharded by
contiguous index ranges, no collective on the data path; the only exchange is the final all-gather of the
fixed-stride proof records (RCCL when the tensors live on the GPU, gloo on CPU in tests).

The reference has no distributed code (SURVEY.md sections 2, 5); ops are independent
(/root/reference/src/advanced/batch.rs:123-131), which is what makes this sharding exact.
"""
import numpy as np

from . import _native


def shard_bounds(n, world, rank):
    """Contiguous balanced split: the first n % world ranks get one extra op."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def process_range_batch_sharded(values, mins, maxs, seeds, prover=None, device=None):
    """Every rank passes the FULL op list (and 32*n seed bytes); returns the full ordered proof list on every rank."""
    import torch
    import torch.distributed as dist

    if prover is None:
        from .api import prove_range_batch as prover
    n = len(values)
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    lo, hi = shard_bounds(n, world, rank)
    seeds = bytes(seeds)
    mine = prover(values[lo:hi], mins[lo:hi], maxs[lo:hi], seeds[32 * lo: 32 * hi]) if hi > lo else []
    if world == 1:
        return mine
    stride = _native.RANGE_PROOF_BYTES
    cap = shard_bounds(n, world, 0)[1]                      # largest shard
    buf = np.zeros((cap, stride), dtype=np.uint8)
    for i, p in enumerate(mine):
        assert len(p) == stride
        buf[i] = np.frombuffer(p, dtype=np.uint8)
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    out = []
    for r in range(world):
        a, b = shard_bounds(n, world, r)
        arr = parts[r].cpu().numpy()
        out.extend(arr[i].tobytes() for i in range(b - a))
    return out
