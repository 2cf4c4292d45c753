"""BASELINE configs at their stated sizes through the C ABI: C3 (4096 prove_equality) and C5 (16 384 mixed ops) on one GPU,
plus the metric's own 4096-op mixed batch.  At these sizes the oracle cannot re-prove everything, so: size-independent
properties (every envelope accepted by the GPU verifiers with its own public values, tampered ones rejected, results
independent of batch composition and of sharding) and a byte-exact sample against oracle/c."""
import ctypes
import os

import numpy as np
import pytest

from libzkp_amd import workloads as wl
from util import P, U64

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def L():
    from libzkp_amd import _native
    import libzkp_amd.api as api
    lib = _native.lib()
    _native.check(lib.zkp_hip_init(0), "zkp_hip_init")
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        blob = open(os.path.join(GOLD, name), "rb").read()
        assert lib.zkp_hip_groth16_load_key(kind, blob, len(blob)) == 0, _native.last_error()
    with api._snark_lock:
        api._keys_loaded[0] = api._keys_loaded[1] = True
        api._reinstall.clear()
    return lib


@pytest.fixture(scope="module")
def orc(oracle_c):
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        pk = open(os.path.join(GOLD, name), "rb").read()
        assert oracle_c.zkp_oracle_g16_load_key(kind, pk, U64(len(pk))) == 0
    return oracle_c


def run(L, ops, lists, seeds):
    from libzkp_amd import _native
    n = len(ops)
    cap = wl.max_output_bytes(ops)
    out = np.zeros(cap, dtype=np.uint8); off = np.zeros(n + 1, dtype=np.uint64); st = np.zeros(n, dtype=np.int32)
    rc = L.zkp_hip_process_batch(n, P(ops), P(lists), P(seeds), P(out), cap, P(off), P(st))
    assert rc == 0, _native.last_error()
    return out, off.astype(np.int64), st


def oracle_sample(orc, ops, lists, seeds, idx):
    """the ops at `idx` re-proved by oracle/c's process_batch port (same seeds) -> list of envelopes"""
    sub = ops[idx].copy()
    sd = np.ascontiguousarray(seeds.reshape(len(ops), 32)[idx]).ravel()
    cap = wl.max_output_bytes(sub)
    out = np.zeros(cap, dtype=np.uint8); off = np.zeros(len(idx) + 1, dtype=np.uint64); st = np.zeros(len(idx), dtype=np.int32)
    assert orc.zkp_oracle_process_batch(U64(len(idx)), P(sub), P(lists), P(sd), P(out), U64(cap), P(off), P(st), 16) == 0
    return [out[int(off[k]):int(off[k + 1])].tobytes() for k in range(len(idx))]


def strided(out, off, idx, stride):
    """envelopes at `idx` as a (len(idx), stride) array + lengths, the layout of the batched verifiers"""
    buf = np.zeros((len(idx), stride), dtype=np.uint8); lens = np.zeros(len(idx), dtype=np.uint32)
    for k, i in enumerate(idx):
        lens[k] = off[i + 1] - off[i]
        buf[k, :lens[k]] = out[off[i]:off[i + 1]]
    return buf, lens


def verify_all(L, ops, lists, out, off):
    """every envelope of a mixed batch through the GPU verifier of its scheme, with the op's own public values"""
    k = ops["kind"]
    ix = np.nonzero(k == wl.OP_RANGE)[0]
    if len(ix):
        buf, lens = strided(out, off, ix, 1478); ok = np.zeros(len(ix), dtype=np.uint8)
        assert L.zkp_hip_verify_range_batch(len(ix), P(buf), 1478, P(lens), P(ops["b"][ix].copy()), P(ops["c"][ix].copy()), P(ok)) == 0 and ok.all()
    ix = np.nonzero(k == wl.OP_EQUALITY)[0]
    if len(ix):
        buf, lens = strided(out, off, ix, 298); ok = np.zeros(len(ix), dtype=np.uint8)
        assert L.zkp_hip_verify_equality_batch(len(ix), P(buf), 298, P(lens), P(ok)) == 0 and ok.all()
        com = np.zeros((len(ix), 32), dtype=np.uint8)
        assert L.zkp_hip_snark_commit_value_batch(len(ix), P(ops["a"][ix].copy()), P(com)) == 0
        assert (buf[:, 266:298] == com).all()                       # the public input is the MiMC commitment of the op's value
    ix = np.nonzero(k == wl.OP_MEMBERSHIP)[0]
    if len(ix):
        stride = wl.membership_bytes(64)
        buf, lens = strided(out, off, ix, stride); ok = np.zeros(len(ix), dtype=np.uint8)
        assert L.zkp_hip_verify_membership_batch(len(ix), P(buf), stride, P(lens), P(ok)) == 0 and ok.all()
        for kk in (0, len(ix) // 2, len(ix) - 1):                   # the embedded set is the op's set
            o = ops[ix[kk]]
            s = lists[int(o["list_off"]):int(o["list_off"]) + int(o["count"])]
            assert buf[kk, 14:14 + 8 * len(s)].tobytes() == s.astype("<u8").tobytes()
    ix = np.nonzero(k == wl.OP_IMPROVEMENT)[0]
    if len(ix):
        buf, lens = strided(out, off, ix, 3527); ok = np.zeros(len(ix), dtype=np.uint8)
        assert L.zkp_hip_verify_improvement_batch(len(ix), P(buf), 3527, P(lens), P(ops["a"][ix].copy()), P(ok)) == 0 and ok.all()


def test_c3_4096_equality(L, orc):
    n = 4096
    ops, lists, seeds = wl.equality_ops(n, 2)
    out, off, st = run(L, ops, lists, seeds)
    assert not st.any() and (np.diff(off) == 298).all()
    verify_all(L, ops, lists, out, off)
    idx = np.arange(7, n, 128)                                      # 32 envelopes byte for byte against oracle/c
    want = oracle_sample(orc, ops, lists, seeds, idx)
    assert all(out[off[i]:off[i + 1]].tobytes() == w for i, w in zip(idx, want))
    # slice independence (a ragged slice re-proved alone gives the same bytes)
    lo, hi = 1000, 1301
    sub, soff, _ = run(L, ops[lo:hi].copy(), lists, seeds[32 * lo:32 * hi].copy())
    assert sub[:soff[-1]].tobytes() == out[off[lo]:off[hi]].tobytes()
    # snark.rs:639-640 / tests/integration.rs:78-85: a wrong commitment and a flipped proof byte are rejected
    buf, lens = strided(out, off, np.arange(64), 298)
    bad = buf.copy(); bad[:, 12] ^= 1
    ok = np.ones(64, dtype=np.uint8)
    assert L.zkp_hip_verify_equality_batch(64, P(bad), 298, P(lens), P(ok)) == 0 and not ok.any()
    bad = buf.copy(); bad[:, 270] ^= 1
    assert L.zkp_hip_verify_equality_batch(64, P(bad), 298, P(lens), P(ok)) == 0 and not ok.any()


def test_metric_batch_4096_mixed(L, orc):
    n = 4096
    ops, lists, seeds = wl.mixed_ops(n, 5)
    out, off, st = run(L, ops, lists, seeds)
    assert not st.any()
    assert (out[off[:-1] + 1] == np.array([1, 2, 4, 5], dtype=np.uint8)[np.arange(n) % 4]).all()       # scheme ids in the caller's order
    verify_all(L, ops, lists, out, off)
    idx = np.arange(0, n, 129)                                      # 32 ops, kinds rotate (129 = 1 mod 4)
    want = oracle_sample(orc, ops, lists, seeds, idx)
    assert sorted(set(int(k) for k in ops["kind"][idx])) == [1, 2, 4, 5]
    assert all(out[off[i]:off[i + 1]].tobytes() == w for i, w in zip(idx, want))


def test_c5_16384_mixed_on_one_gpu_and_two_shards(L, orc):
    from libzkp_amd import _native
    n = 16384
    ops, lists, seeds = wl.mixed_ops(n, 5)
    out, off, st = run(L, ops, lists, seeds)
    assert not st.any()
    assert (out[off[:-1] + 1] == np.array([1, 2, 4, 5], dtype=np.uint8)[np.arange(n) % 4]).all()
    verify_all(L, ops, lists, out, off)
    idx = np.arange(3, n, 257)                                      # 64 ops, kinds rotate (257 = 1 mod 4)
    want = oracle_sample(orc, ops, lists, seeds, idx)
    assert all(out[off[i]:off[i + 1]].tobytes() == w for i, w in zip(idx, want))
    total = int(off[-1])
    # the same batch with this GPU registered as two shards (one process, two host workers): identical bytes
    L.zkp_hip_shutdown()
    try:
        _native.init_devices([0, 0])
        for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
            blob = open(os.path.join(GOLD, name), "rb").read()
            assert L.zkp_hip_groth16_load_key(kind, blob, len(blob)) == 0
        out2, off2, st2 = run(L, ops, lists, seeds)
        assert (off2 == off).all() and not st2.any() and out2[:total].tobytes() == out[:total].tobytes()
    finally:
        L.zkp_hip_shutdown()
        _native.check(L.zkp_hip_init(0), "zkp_hip_init")
        for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
            blob = open(os.path.join(GOLD, name), "rb").read()
            assert L.zkp_hip_groth16_load_key(kind, blob, len(blob)) == 0
