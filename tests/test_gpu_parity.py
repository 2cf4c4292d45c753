"""Parity of the HIP path (through the C ABI) with the oracle on an MI355X.  Bit-exact: integer/byte work."""
import ctypes

import numpy as np
import pytest

from util import P, PROOF, U64, oracle_prove, oracle_verify, outputs, workload

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from libzkp_amd import _native
    L = _native.lib()                       # raises if the extension is missing: no fallback
    _native.check(L.zkp_hip_init(0), "zkp_hip_init")
    return L


def hip_prove(L, v, mn, mx, seeds, stride=PROOF):
    n = len(v)
    out, lens, st = outputs(n, stride)
    rc = L.zkp_hip_prove_range_batch(n, P(v), P(mn), P(mx), 64, None if seeds is None else P(seeds), P(out), stride, P(lens), P(st))
    return rc, out, lens, st


@pytest.mark.parametrize("n", [1, 3, 64, 257, 1000])
def test_bit_exact_vs_oracle(hip, oracle_c, n):
    v, mn, mx, seeds = workload(n, 100 + n)
    rc, out, lens, st = hip_prove(hip, v, mn, mx, seeds)
    rc2, o2, l2, s2 = oracle_prove(oracle_c, v, mn, mx, seeds, threads=16)
    assert rc == 0 and rc2 == 0 and (lens == PROOF).all() and (st == 0).all()
    assert (out == o2).all()


def test_golden_vectors(hip, golden_bp):
    for c in golden_bp["range"]:
        g = lambda x: np.array([x], dtype=np.uint64)  # noqa: E731
        sd = np.frombuffer(bytes.fromhex(c["seed"]), dtype=np.uint8).copy()
        rc, out, lens, st = hip_prove(hip, g(c["value"]), g(c["min"]), g(c["max"]), sd)
        assert rc == 0 and out[0].tobytes().hex() == c["proof"]


def test_edge_values(hip, oracle_c):
    """value at either bound, degenerate range, full 64-bit range, mixed bounds, ragged stride."""
    v = np.array([0, 2**32, 7, 0, 2**64 - 1, 2**63, 1], dtype=np.uint64)
    mn = np.array([0, 0, 7, 0, 0, 2**63, 1], dtype=np.uint64)
    mx = np.array([2**32, 2**32, 7, 2**64 - 1, 2**64 - 1, 2**63, 2], dtype=np.uint64)
    seeds = np.arange(32 * len(v), dtype=np.uint32).astype(np.uint8)
    for stride in (PROOF, 1500, 2048):
        rc, out, lens, st = hip_prove(hip, v, mn, mx, seeds, stride)
        rc2, o2, l2, s2 = oracle_prove(oracle_c, v, mn, mx, seeds, threads=8, stride=stride)
        assert rc == 0 and (out[:, :PROOF] == o2[:, :PROOF]).all() and (out[:, PROOF:] == 0).all()
        allok, ok = oracle_verify(oracle_c, out, lens, mn, mx)
        assert allok == 1


def test_invalid_ops_fail_per_item_and_leave_no_bytes(hip, oracle_c):
    v = np.array([5, 11, 5, 3], dtype=np.uint64)
    mn = np.array([0, 0, 10, 0], dtype=np.uint64)
    mx = np.array([10, 10, 0, 3], dtype=np.uint64)
    seeds = np.zeros(32 * 4, dtype=np.uint8)
    rc, out, lens, st = hip_prove(hip, v, mn, mx, seeds)
    assert rc == 1 and list(st) == [0, 1, 1, 0] and list(lens) == [PROOF, 0, 0, PROOF]
    assert (out[1] == 0).all() and (out[2] == 0).all()
    ok_idx = [0, 3]
    rc2, o2, _, _ = oracle_prove(oracle_c, v[ok_idx], mn[ok_idx], mx[ok_idx], seeds[:64].copy(), threads=2)
    # seeds are all-zero so op 3 shares op 0's seed bytes
    assert (out[0] == o2[0]).all() and (out[3] == o2[1]).all()


def test_empty_batch_and_bad_arguments(hip):
    from libzkp_amd import _native
    z = np.zeros(0, dtype=np.uint64)
    out, lens, st = outputs(1)
    assert hip.zkp_hip_prove_range_batch(0, P(z), P(z), P(z), 64, None, P(out), PROOF, P(lens), P(st)) == 0
    one = np.ones(1, dtype=np.uint64)
    assert hip.zkp_hip_prove_range_batch(1, P(one), P(one), P(one), 12, None, P(out), PROOF, P(lens), P(st)) == -2
    assert "n_bits" in _native.last_error()
    assert hip.zkp_hip_prove_range_batch(1, P(one), P(one), P(one), 64, None, P(out), 100, P(lens), P(st)) == -3


def test_os_randomness_when_no_seeds(hip, oracle_c):
    v, mn, mx, _ = workload(16, 5)
    rc, a, la, _ = hip_prove(hip, v, mn, mx, None)
    rc2, b, lb, _ = hip_prove(hip, v, mn, mx, None)
    assert rc == 0 and rc2 == 0 and not (a == b).all()
    assert oracle_verify(oracle_c, a, la, mn, mx)[0] == 1 and oracle_verify(oracle_c, b, lb, mn, mx)[0] == 1


def test_full_size_batch_properties(hip, oracle_c):
    """BASELINE configs[1]: 4096 prove_range(v, 0, 2^32).  Size-independent properties: every proof is accepted by
    the restated verifier with its own bounds, rejected with a tighter bound or a flipped byte; results do not depend
    on batch composition (a slice re-proved alone gives the same bytes)."""
    n = 4096
    v, mn, mx, seeds = workload(n, 1)
    rc, out, lens, st = hip_prove(hip, v, mn, mx, seeds)
    assert rc == 0 and (lens == PROOF).all()
    allok, ok = oracle_verify(oracle_c, out, lens, mn, mx, threads=16)
    assert allok == 1
    # slice independence (also exercises a ragged, non-multiple-of-256 batch)
    lo, hi = 1000, 1301
    rc, sub, _, _ = hip_prove(hip, v[lo:hi].copy(), mn[lo:hi].copy(), mx[lo:hi].copy(), seeds[32 * lo: 32 * hi].copy())
    assert rc == 0 and (sub == out[lo:hi]).all()
    # oracle prover on a sample
    idx = np.arange(0, n, 64)
    rc2, o2, _, _ = oracle_prove(oracle_c, v[idx], mn[idx], mx[idx], seeds.reshape(n, 32)[idx].ravel().copy(), threads=16)
    assert (out[idx] == o2).all()
    # negative cases (integration.rs:78-85, bulletproofs.rs:704)
    bad = out[:64].copy()
    bad[:, 12] ^= 1
    assert oracle_verify(oracle_c, bad, lens[:64], mn[:64], mx[:64])[1].sum() == 0
    tight = v[:64].copy() - 1
    sel = v[:64] > 0
    okt = oracle_verify(oracle_c, out[:64], lens[:64], mn[:64], tight)[1]
    assert okt[sel].sum() == 0


def test_device_resident_entry_point(hip, oracle_c):
    import torch
    n = 300
    v, mn, mx, seeds = workload(n, 77)
    dev = torch.device("cuda", 0)
    tv, tmn, tmx = (torch.from_numpy(a.view(np.int64)).to(dev) for a in (v, mn, mx))
    ts = torch.from_numpy(seeds).to(dev)
    tout = torch.zeros((n, PROOF), dtype=torch.uint8, device=dev)
    tlen = torch.zeros(n, dtype=torch.int32, device=dev)
    tst = torch.zeros(n, dtype=torch.int32, device=dev)
    any_failed = ctypes.c_int(-1)
    rc = hip.zkp_hip_prove_range_batch_device(n, tv.data_ptr(), tmn.data_ptr(), tmx.data_ptr(), 64, ts.data_ptr(), tout.data_ptr(), PROOF,
                                              tlen.data_ptr(), tst.data_ptr(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream),
                                              ctypes.byref(any_failed))
    torch.cuda.synchronize()
    assert rc == 0 and any_failed.value == 0
    rc2, o2, _, _ = oracle_prove(oracle_c, v, mn, mx, seeds, threads=16)
    assert (tout.cpu().numpy() == o2).all()


def test_python_api_on_gpu(hip, oracle_c):
    import libzkp_amd as z
    p = z.prove_range(50, 0, 100)
    assert len(p) == PROOF and p[0] == 2 and p[1] == 1
    assert oracle_c.zkp_oracle_verify_range(p, len(p), U64(0), U64(100)) == 1
    assert oracle_c.zkp_oracle_verify_range(p, len(p), U64(0), U64(49)) == 0
    b = z.create_proof_batch()
    for i in range(5):
        z.batch_add_range_proof(b, 10 * i, 0, 100 + i)
    proofs = z.process_batch(b)
    assert len(proofs) == 5
    for i, p in enumerate(proofs):       # order preserved (batch.rs:123-131)
        assert oracle_c.zkp_oracle_verify_range(p, len(p), U64(0), U64(100 + i)) == 1
    with pytest.raises(ValueError):
        z.process_batch(b)
    r = z.benchmark_proof_generation("range", 3)    # BASELINE configs[0] shape
    assert r["proof_type"] == "range" and float(r["success_rate"]) == 100.0 and float(r["proofs_per_second"]) > 0


def _flat(lists):
    return np.array([x for l in lists for x in l], dtype=np.uint64), np.array([len(l) for l in lists], dtype=np.uint32)


def test_threshold_bit_exact_vs_oracle(hip, oracle_c, golden_bp):
    """bulletproofs.rs:309-366 framing (762 bytes), incl. the benchmark harness inputs (mod.rs:96) and invalid ops."""
    lists = [[10, 20, 30, 40], [2**64 - 1], [0], [5, 6, 7] * 20, [1, 2], [2**63, 2**63]]
    thr = np.array([50, 2**64 - 1, 0, 100, 4, 1], dtype=np.uint64)     # op 4: sum < threshold; op 5: overflow
    flat, counts = _flat(lists)
    n = len(lists)
    seeds = np.arange(32 * n, dtype=np.uint32).astype(np.uint8)
    out, lens, st = outputs(n, 800)
    rc = hip.zkp_hip_prove_threshold_batch(n, P(flat), P(counts), P(thr), 64, P(seeds), P(out), 800, P(lens), P(st))
    assert rc == 1 and list(st) == [0, 0, 0, 0, 1, 1] and list(lens) == [762, 762, 762, 762, 0, 0]
    ref = ctypes.create_string_buffer(1024)
    ol = ctypes.c_uint32()
    for i in range(4):
        vals = (ctypes.c_uint64 * len(lists[i]))(*lists[i])
        assert oracle_c.zkp_oracle_prove_threshold(vals, len(lists[i]), U64(int(thr[i])), 64, seeds[32 * i: 32 * i + 32].tobytes(), ref, 1024, ctypes.byref(ol)) == 0
        assert out[i, :762].tobytes() == ref.raw[:762] and (out[i, 762:] == 0).all()
        assert oracle_c.zkp_oracle_verify_threshold(out[i, :762].tobytes(), 762, U64(int(thr[i]))) == 1
    assert (out[4] == 0).all() and (out[5] == 0).all()
    c = golden_bp["threshold"][0]
    flat, counts = _flat([c["values"]])
    sd = np.frombuffer(bytes.fromhex(c["seed"]), dtype=np.uint8).copy()
    out, lens, st = outputs(1, 762)
    assert hip.zkp_hip_prove_threshold_batch(1, P(flat), P(counts), P(np.array([c["threshold"]], dtype=np.uint64)), 64, P(sd), P(out), 762, P(lens), P(st)) == 0
    assert out[0].tobytes().hex() == c["proof"]


def test_consistency_bit_exact_vs_oracle(hip, oracle_c, golden_bp):
    """bulletproofs.rs:368-437: k commitments, k-1 range proofs with blinding differences, SHA-256 commitment digest."""
    lists = [[10, 20, 30, 40, 50], [7], [0, 0, 2**64 - 1], [3, 2], [1, 1, 1, 1, 1, 1, 1, 1]]
    flat, counts = _flat(lists)
    n = len(lists)
    stride = max(int(hip.zkp_hip_consistency_proof_bytes(len(l))) for l in lists)
    assert int(hip.zkp_hip_consistency_proof_bytes(3)) == 1558
    seeds = (np.arange(32 * n, dtype=np.uint32) * 7 + 1).astype(np.uint8)
    out, lens, st = outputs(n, stride)
    rc = hip.zkp_hip_prove_consistency_batch(n, P(flat), P(counts), P(seeds), P(out), stride, P(lens), P(st))
    assert rc == 1 and list(st) == [0, 0, 0, 1, 0]
    ref = ctypes.create_string_buffer(stride + 16)
    ol = ctypes.c_uint32()
    for i in (0, 1, 2, 4):
        d = (ctypes.c_uint64 * len(lists[i]))(*lists[i])
        assert oracle_c.zkp_oracle_prove_consistency(d, len(lists[i]), seeds[32 * i: 32 * i + 32].tobytes(), ref, stride + 16, ctypes.byref(ol)) == 0
        assert lens[i] == ol.value and out[i, : lens[i]].tobytes() == ref.raw[: ol.value]
        assert oracle_c.zkp_oracle_verify_consistency(out[i, : lens[i]].tobytes(), int(lens[i])) == 1
    c = golden_bp["consistency"][0]
    flat, counts = _flat([c["data"]])
    sd = np.frombuffer(bytes.fromhex(c["seed"]), dtype=np.uint8).copy()
    sz = len(c["proof"]) // 2
    out, lens, st = outputs(1, sz)
    assert hip.zkp_hip_prove_consistency_batch(1, P(flat), P(counts), P(sd), P(out), sz, P(lens), P(st)) == 0
    assert out[0].tobytes().hex() == c["proof"]


def test_mixed_process_batch_keeps_order(hip, oracle_c):
    """examples/demo.rs:66-105 shape restricted to the Bulletproofs-backed variants: mixed ops, order preserved."""
    import libzkp_amd as z
    b = z.create_proof_batch()
    z.batch_add_range_proof(b, 25, 18, 65)
    z.batch_add_threshold_proof(b, [100, 200, 300], 500)
    z.batch_add_consistency_proof(b, [10, 20, 30])
    z.batch_add_range_proof(b, 7, 0, 10)
    z.batch_add_threshold_proof(b, [1], 1)
    seeds = bytes(range(160))
    proofs = z.process_batch(b, seeds=seeds)
    assert [p[1] for p in proofs] == [1, 3, 6, 1, 3]
    assert oracle_c.zkp_oracle_verify_range(proofs[0], len(proofs[0]), U64(18), U64(65)) == 1
    assert oracle_c.zkp_oracle_verify_threshold(proofs[1], len(proofs[1]), U64(500)) == 1
    assert oracle_c.zkp_oracle_verify_consistency(proofs[2], len(proofs[2])) == 1
    assert oracle_c.zkp_oracle_verify_range(proofs[3], len(proofs[3]), U64(0), U64(10)) == 1
    assert oracle_c.zkp_oracle_verify_threshold(proofs[4], len(proofs[4]), U64(1)) == 1
    # per-op seeds follow the op, not its position inside a variant bucket
    ref = ctypes.create_string_buffer(2048)
    ol = ctypes.c_uint32()
    assert oracle_c.zkp_oracle_prove_range(U64(7), U64(0), U64(10), 64, seeds[96:128], ref, 2048, ctypes.byref(ol)) == 0
    assert proofs[3] == ref.raw[: ol.value]
    for t in ("threshold", "consistency"):
        r = z.benchmark_proof_generation(t, 2)
        assert r["proof_type"] == t and float(r["success_rate"]) == 100.0
