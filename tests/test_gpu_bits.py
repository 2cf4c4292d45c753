"""prove_range_with_bits / prove_threshold_with_bits on the MI355X (range_proof.rs:14-27,
threshold_proof.rs:17-32, bulletproofs.rs:112-178,309-366): 8/16/32-bit proofs are byte-identical to the oracle's, verify on
the GPU (widths mixed in one batch) and in the oracle, and 64-bit output is unchanged after narrower batches."""
import numpy as np
import pytest

from util import P, U64, oracle_prove, oracle_verify, outputs, workload

pytestmark = pytest.mark.gpu

SIZE = {8: 1094, 16: 1222, 32: 1350, 64: 1478}
TSIZE = {8: 570, 16: 634, 32: 698, 64: 762}


@pytest.fixture(scope="module")
def hip():
    from libzkp_amd import _native
    L = _native.lib()
    _native.check(L.zkp_hip_init(0), "zkp_hip_init")
    return L


def test_sizes(hip):
    for bits in SIZE:
        assert hip.zkp_hip_range_proof_bytes(bits) == SIZE[bits] and hip.zkp_hip_threshold_proof_bytes(bits) == TSIZE[bits]
    assert hip.zkp_hip_range_proof_bytes(12) == 0 and hip.zkp_hip_threshold_proof_bytes(0) == 0


def test_range_widths_equal_oracle_and_verify(hip, oracle_c):
    rng = np.random.default_rng(8)
    rows, lens_all, mns, mxs = [], [], [], []
    for bits in (8, 32, 16, 64, 8):                                     # families are built lazily; 64 again after narrower ones
        n = 257
        cap = 2**bits - 1
        span = rng.integers(0, min(cap, 2**62), n, dtype=np.uint64, endpoint=True)
        mn = rng.integers(0, 2**40, n, dtype=np.uint64)
        mx = mn + span
        v = mn + (rng.integers(0, 2**62, n, dtype=np.uint64) % (span + np.uint64(1)))
        v[0], mn[0], mx[0] = cap, 0, cap                                # both differences at the width's limits
        seeds = np.frombuffer(rng.bytes(32 * n), dtype=np.uint8).copy()
        out, lens, st = outputs(n)
        assert hip.zkp_hip_prove_range_batch(n, P(v), P(mn), P(mx), bits, P(seeds), P(out), 1478, P(lens), P(st)) == 0
        o2, l2, s2 = outputs(n)
        assert oracle_c.zkp_oracle_prove_range_batch(U64(n), P(v), P(mn), P(mx), bits, P(seeds), P(o2), U64(1478), P(l2), P(s2), 16) == 0
        assert (lens == SIZE[bits]).all() and (l2 == lens).all()
        assert (out == o2).all()
        assert oracle_verify(oracle_c, out, lens, mn, mx, threads=16)[1].all()
        rows.append(out); lens_all.append(lens); mns.append(mn); mxs.append(mx)
    out, lens, mn, mx = np.concatenate(rows), np.concatenate(lens_all), np.concatenate(mns), np.concatenate(mxs)
    n = len(lens)
    ok = np.zeros(n, dtype=np.uint8)
    assert hip.zkp_hip_verify_range_batch(n, P(out), 1478, P(lens), P(mn), P(mx), P(ok)) == 0
    assert (ok == 1).all()
    t = out.copy()
    pos = rng.integers(0, 1094, n)
    t[np.arange(n), pos] ^= (1 << rng.integers(0, 8, n)).astype(np.uint8)
    t[3, 26] = 16; t[3, 27:30] = 0                                        # another valid width than the proof was made for
    assert hip.zkp_hip_verify_range_batch(n, P(t), 1478, P(lens), P(mn), P(mx), P(ok)) == 0
    want = oracle_verify(oracle_c, t, lens, mn, mx, threads=16)[1]
    assert (ok == want).all() and want.sum() == 0


def test_width_capacity_and_bad_width(hip):
    v = np.array([300, 5, 255], dtype=np.uint64); mn = np.array([0, 0, 0], dtype=np.uint64); mx = np.array([310, 300, 255], dtype=np.uint64)
    seeds = np.zeros(96, dtype=np.uint8)
    out, lens, st = outputs(3)
    assert hip.zkp_hip_prove_range_batch(3, P(v), P(mn), P(mx), 8, P(seeds), P(out), 1478, P(lens), P(st)) == 1
    assert list(st) == [1, 1, 0] and list(lens) == [0, 0, 1094] and not out[:2].any()
    assert hip.zkp_hip_prove_range_batch(3, P(v), P(mn), P(mx), 12, P(seeds), P(out), 1478, P(lens), P(st)) == -2
    assert hip.zkp_hip_prove_range_batch(3, P(v), P(mn), P(mx), 8, P(seeds), P(out), 1000, P(lens), P(st)) == -3     # stride < 1094


def test_threshold_widths(hip, oracle_c):
    import ctypes
    rng = np.random.default_rng(3)
    for bits in (8, 16, 32):
        n = 65
        cap = 2**bits - 1
        counts = rng.integers(1, 5, n).astype(np.uint32)
        lists = [rng.integers(0, 2**30, int(c), dtype=np.uint64) for c in counts]
        thr = np.array([int(x.sum()) - int(rng.integers(0, min(cap, int(x.sum())), endpoint=True)) for x in lists], dtype=np.uint64)
        flat = np.concatenate(lists)
        seeds = np.frombuffer(rng.bytes(32 * n), dtype=np.uint8).copy()
        out = np.zeros((n, 762), dtype=np.uint8); lens = np.zeros(n, dtype=np.uint32); st = np.zeros(n, dtype=np.int32)
        assert hip.zkp_hip_prove_threshold_batch(n, P(flat), P(counts), P(thr), bits, P(seeds), P(out), 762, P(lens), P(st)) == 0
        assert (lens == TSIZE[bits]).all()
        for i in range(n):
            buf = ctypes.create_string_buffer(1024); ln = ctypes.c_uint32()
            vals = (ctypes.c_uint64 * len(lists[i]))(*[int(x) for x in lists[i]])
            rc = oracle_c.zkp_oracle_prove_threshold(vals, len(lists[i]), U64(int(thr[i])), bits, seeds[32 * i: 32 * i + 32].tobytes(), buf, 1024, ctypes.byref(ln))
            assert rc == 0 and ln.value == TSIZE[bits]
            assert buf.raw[: ln.value] == out[i, : ln.value].tobytes()
        ok = np.zeros(n, dtype=np.uint8)
        assert hip.zkp_hip_verify_threshold_batch(n, P(out), 762, P(lens), P(thr), P(ok)) == 0
        assert (ok == 1).all()
        assert hip.zkp_hip_verify_threshold_batch(n, P(out), 762, P(lens), P(thr + np.uint64(1)), P(ok)) == 0
        assert (ok == 0).all()
    # sum - threshold does not fit in 8 bits
    flat = np.array([1000], dtype=np.uint64); counts = np.array([1], dtype=np.uint32); thr = np.array([10], dtype=np.uint64)
    out = np.zeros((1, 762), dtype=np.uint8); lens = np.zeros(1, dtype=np.uint32); st = np.zeros(1, dtype=np.int32)
    assert hip.zkp_hip_prove_threshold_batch(1, P(flat), P(counts), P(thr), 8, P(np.zeros(32, dtype=np.uint8)), P(out), 762, P(lens), P(st)) == 1
    assert st[0] == 1 and lens[0] == 0


def test_python_mirror(hip, oracle_c):
    import libzkp_amd as z
    p8 = z.prove_range_with_bits(200, 100, 300, 8)
    assert len(p8) == 1094 and z.verify_range(p8, 100, 300) and not z.verify_range(p8, 100, 301)
    p64 = z.prove_range(200, 100, 300)
    assert len(p64) == 1478 and z.verify_range(p64, 100, 300)
    with pytest.raises(z.ZkpBackendError, match="exceeds 8-bit capacity"):
        z.prove_range_with_bits(500, 100, 600, 8)
    with pytest.raises(z.ZkpBackendError, match="n_bits"):
        z.prove_range_with_bits(5, 0, 9, 7)
    t16 = z.prove_threshold_with_bits([10, 20, 30], 50, 16)
    assert len(t16) == 634 and z.verify_threshold(t16, 50) and not z.verify_threshold(t16, 51)
    with pytest.raises(z.ZkpBackendError, match="exceeds 8-bit capacity"):
        z.prove_threshold_with_bits([1000], 10, 8)
    info = z.get_proof_info(p8)
    assert info["scheme"] == 1
