"""Synthetic workloads of SURVEY.md section 8(d) as C-ABI operation arrays (shared by bench.py, the tests and tools/).

`OP_DTYPE` is `zkp_hip_op` of include/libzkp_hip.h as a numpy structured type, so a batch is one contiguous array that
goes to `zkp_hip_process_batch` / `zkp_hip_batch_stage` without per-op Python marshalling.
"""
import hashlib

import numpy as np

OP_RANGE, OP_EQUALITY, OP_THRESHOLD, OP_MEMBERSHIP, OP_IMPROVEMENT, OP_CONSISTENCY = 1, 2, 3, 4, 5, 6
OP_DTYPE = np.dtype([("kind", "<u4"), ("count", "<u4"), ("a", "<u8"), ("b", "<u8"), ("c", "<u8"), ("list_off", "<u8")])
assert OP_DTYPE.itemsize == 40

RANGE_BYTES = 1478
EQUALITY_BYTES = 298
IMPROVEMENT_MAX_BYTES = 3527


def membership_bytes(set_len):
    return 10 + 4 + 8 * set_len + 256 + 32


def op_seeds(seed, n):
    """Per-proof randomness seeds = SHA-256(seed || i), both as u64 little endian."""
    return np.frombuffer(b"".join(hashlib.sha256(seed.to_bytes(8, "little") + i.to_bytes(8, "little")).digest() for i in range(n)),
                         dtype=np.uint8).copy()


def range_ops(n, seed=1):
    """C2: value ~ U[0, 2^32], min = 0, max = 2^32."""
    rng = np.random.default_rng(seed)
    ops = np.zeros(n, dtype=OP_DTYPE)
    ops["kind"] = OP_RANGE
    ops["a"] = rng.integers(0, 2**32, n, dtype=np.uint64, endpoint=True)
    ops["c"] = 2**32
    return ops, np.zeros(1, dtype=np.uint64), op_seeds(seed, n)


def equality_ops(n, seed=2):
    """C3: a = b ~ U[0, 2^64)."""
    rng = np.random.default_rng(seed)
    ops = np.zeros(n, dtype=OP_DTYPE)
    ops["kind"] = OP_EQUALITY
    ops["a"] = rng.integers(0, 2**64, n, dtype=np.uint64)
    ops["b"] = ops["a"]
    return ops, np.zeros(1, dtype=np.uint64), op_seeds(seed, n)


def improvement_ops(n, seed=3):
    """C4 at the reference's real parameters: old ~ U[0, 2^63), new = old + 1 + U[0, 2^32)."""
    rng = np.random.default_rng(seed)
    ops = np.zeros(n, dtype=OP_DTYPE)
    ops["kind"] = OP_IMPROVEMENT
    ops["a"] = rng.integers(0, 2**63, n, dtype=np.uint64)
    ops["b"] = ops["a"] + np.uint64(1) + rng.integers(0, 2**32, n, dtype=np.uint64)
    return ops, np.zeros(1, dtype=np.uint64), op_seeds(seed, n)


def mixed_ops(n, seed=5, set_len=16):
    """C5's mix: op i is range / equality / membership(set_len) / improvement for i mod 4 = 0 / 1 / 2 / 3, interleaved;
    membership sets are `set_len` distinct values of U[0, 2^32) with value = set[i mod set_len]."""
    rng = np.random.default_rng(seed)
    ops = np.zeros(n, dtype=OP_DTYPE)
    k = np.arange(n) % 4
    r, e, m, s = (k == 0), (k == 1), (k == 2), (k == 3)
    ops["kind"][r] = OP_RANGE
    ops["a"][r] = rng.integers(0, 2**32, int(r.sum()), dtype=np.uint64, endpoint=True)
    ops["c"][r] = 2**32
    ops["kind"][e] = OP_EQUALITY
    ops["a"][e] = rng.integers(0, 2**64, int(e.sum()), dtype=np.uint64)
    ops["b"][e] = ops["a"][e]
    nm = int(m.sum())
    sets = np.zeros((nm, set_len), dtype=np.uint64)
    for j in range(nm):
        sets[j] = rng.choice(2**32, set_len, replace=False)
    mi = np.nonzero(m)[0]
    ops["kind"][m] = OP_MEMBERSHIP
    ops["count"][m] = set_len
    ops["list_off"][m] = np.arange(nm, dtype=np.uint64) * np.uint64(set_len)
    ops["a"][m] = sets[np.arange(nm), mi % set_len]
    ops["kind"][s] = OP_IMPROVEMENT
    olds = rng.integers(0, 2**63, int(s.sum()), dtype=np.uint64)
    ops["a"][s] = olds
    ops["b"][s] = olds + np.uint64(1) + rng.integers(0, 2**32, int(s.sum()), dtype=np.uint64)
    lists = sets.ravel().copy() if nm else np.zeros(1, dtype=np.uint64)
    return ops, lists, op_seeds(seed, n)


def max_output_bytes(ops):
    """Upper bound of the proof bytes of a batch (improvement envelopes have a data-dependent length)."""
    kind, cnt = ops["kind"], ops["count"].astype(np.int64)
    total = int((kind == OP_RANGE).sum()) * RANGE_BYTES + int((kind == OP_EQUALITY).sum()) * EQUALITY_BYTES
    total += int((kind == OP_IMPROVEMENT).sum()) * IMPROVEMENT_MAX_BYTES + int((kind == OP_THRESHOLD).sum()) * 762
    total += int((10 + 4 + 8 * cnt[kind == OP_MEMBERSHIP] + 256 + 32).sum())
    c = cnt[kind == OP_CONSISTENCY]
    total += int((10 + 4 + 32 * c + (4 + 672 + 32) * np.maximum(c - 1, 0) + 32).sum())
    return total
