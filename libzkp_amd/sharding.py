"""Multi-GPU form of process_batch: one process per GPU, independent ops sharded by contiguous index ranges (per
variant, so every GPU gets the same mix of kernels), no collective on the data path; the only exchange is the final
all-gather of the proof records (RCCL when the tensors live on the GPU, gloo on CPU in tests).

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


def share_snark_keys(kinds, device=None, export=None, install=None):
    """One trusted setup for the whole job: rank 0 loads or generates the proving key of each circuit in
    `kinds` (0 equality, 1 membership) and broadcasts its ark-serialized bytes; the other ranks install exactly that key.
    Without this every rank would run its own random setup and the job's proofs would verify under different keys."""
    import torch
    import torch.distributed as dist

    from . import api
    export = export or api.export_proving_key
    install = install or api.install_proving_key
    if not dist.is_initialized() or dist.get_world_size() == 1:
        for k in kinds:
            export(k)
        return
    rank = dist.get_rank()
    for k in sorted(set(kinds)):
        blob = export(k) if rank == 0 else b""
        size = torch.tensor([len(blob)], dtype=torch.int64)
        if device is not None:
            size = size.to(device)
        dist.broadcast(size, src=0)
        buf = torch.zeros(int(size.item()), dtype=torch.uint8)
        if rank == 0:
            buf = torch.frombuffer(bytearray(blob), dtype=torch.uint8).clone()
        if device is not None:
            buf = buf.to(device)
        dist.broadcast(buf, src=0)
        if rank != 0:
            install(k, buf.cpu().numpy().tobytes())


def process_ops_sharded(ops, seeds=None, prover=None, device=None):
    """Mixed batch (tuples as stored by api.batch_add_*): every rank passes the FULL op list (and 32 bytes of seed per
    op, required when world > 1 so that all ranks agree on the randomness); each variant's bucket is split into
    contiguous slices, one per rank; returns the full ordered proof list on every rank.  Proof records
    have variable length (improvement proofs, membership sets), so sizes are gathered first, then padded payloads."""
    import torch
    import torch.distributed as dist

    from . import api
    if prover is None:
        prover = api.prove_kind
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if world > 1 and seeds is None:
        raise ValueError("sharded proving needs explicit per-op seeds")
    if prover is api.prove_kind:
        share_snark_keys([0] * any(o[0] == "equality" for o in ops) + [1] * any(o[0] == "membership" for o in ops), device=device)

    def select_for(r):
        def select(kind, idx):
            lo, hi = shard_bounds(len(idx), world, r)
            return idx[lo:hi]
        return select

    mine = api.prove_ops(ops, seeds, prover=prover, select=select_for(rank))
    if world == 1:
        return mine
    # which ops each rank owns (deterministic from the op list alone)
    owned = []
    for r in range(world):
        sel = select_for(r)
        own = []
        for kind in api.KINDS:
            own.extend(sel(kind, [i for i, o in enumerate(ops) if o[0] == kind]))
        owned.append(own)
    payload = b"".join(len(mine[i]).to_bytes(4, "little") + mine[i] for i in owned[rank])
    size = torch.tensor([len(payload)], dtype=torch.int64)
    if device is not None:
        size = size.to(device)
    sizes = [torch.zeros_like(size) for _ in range(world)]
    dist.all_gather(sizes, size)
    cap = max(int(x.item()) for x in sizes)
    buf = np.zeros(max(cap, 1), dtype=np.uint8)
    buf[:len(payload)] = np.frombuffer(payload, dtype=np.uint8)
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    out = [None] * len(ops)
    for r in range(world):
        raw = parts[r].cpu().numpy().tobytes()
        pos = 0
        for i in owned[r]:
            ln = int.from_bytes(raw[pos:pos + 4], "little")
            out[i] = raw[pos + 4:pos + 4 + ln]
            pos += 4 + ln
    return out
