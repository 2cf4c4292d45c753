import ctypes

import numpy as np

U64 = ctypes.c_uint64
PROOF = 1478


def P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def workload(n, seed, lo=0, hi=2**32):
    rng = np.random.default_rng(seed)
    v = rng.integers(lo, hi, n, dtype=np.uint64, endpoint=True)
    mn = np.full(n, lo, dtype=np.uint64)
    mx = np.full(n, hi, dtype=np.uint64)
    seeds = rng.integers(0, 256, 32 * n, dtype=np.uint8)
    return v, mn, mx, seeds


def outputs(n, stride=PROOF):
    return np.zeros((n, stride), dtype=np.uint8), np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.int32)


def oracle_prove(lib, v, mn, mx, seeds, threads=8, stride=PROOF):
    n = len(v)
    out, lens, st = outputs(n, stride)
    rc = lib.zkp_oracle_prove_range_batch(U64(n), P(v), P(mn), P(mx), 64, P(seeds), P(out), U64(stride), P(lens), P(st), threads)
    return rc, out, lens, st


def oracle_verify(lib, out, lens, mn, mx, threads=8):
    n = len(lens)
    ok = np.zeros(n, dtype=np.uint8)
    allok = lib.zkp_oracle_verify_range_batch(U64(n), P(out), U64(out.shape[1]), P(lens), P(mn), P(mx), P(ok), threads)
    return allok, ok
