"""Improvement proofs (STARK) on the MI355X through the C ABI against the oracle: bit-exact envelopes, the committed
oracle vectors, verifier acceptance at batch scale, the reference's error behaviour, and the device-pointer entry."""
import ctypes
import hashlib
import json
import os

import numpy as np
import pytest

from oracle.py import stark as s
from util import P

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def hip():
    from libzkp_amd import _native
    L = _native.lib()
    _native.check(L.zkp_hip_init(0), "zkp_hip_init")
    return L


def run(L, olds, news):
    n = len(olds)
    o, w = np.array(olds, dtype=np.uint64), np.array(news, dtype=np.uint64)
    stride = int(L.zkp_hip_improvement_max_bytes())
    out = np.zeros((n, stride), dtype=np.uint8)
    ln, st = np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.int32)
    rc = L.zkp_hip_prove_improvement_batch(n, P(o), P(w), P(out), stride, P(ln), P(st))
    return rc, [out[i, :ln[i]].tobytes() for i in range(n)], st


def test_bit_exact_against_oracle(hip):
    rng = np.random.default_rng(3)
    olds = [0, 30, 5, 0, 2**64 - 2] + [int(x) for x in rng.integers(0, 2**63, 59, dtype=np.uint64)]
    news = [1, 50, 2**63, 2**64 - 1, 2**64 - 1] + [int(o) + 1 + int(d) for o, d in zip(olds[5:], rng.integers(0, 2**32, 59, dtype=np.uint64))]
    rc, proofs, st = run(hip, olds, news)
    assert rc == 0 and (st == 0).all()
    for o, w, p in zip(olds, news, proofs):
        assert p == s.prove_improvement(o, w), (o, w)


def test_committed_vectors(hip):
    gold = json.load(open(os.path.join(GOLD, "stark_oracle_vectors.json")))["vectors"]
    rc, proofs, st = run(hip, [int(v["old"]) for v in gold], [int(v["new"]) for v in gold])
    assert rc == 0
    for v, p in zip(gold, proofs):
        assert len(p) == v["len"] and hashlib.sha256(p).hexdigest() == v["sha256"]


def test_config_c4_batch_verifies(hip):
    """BASELINE config 4 shape: 1024 improvement proofs (old ~ U[0,2^63), new = old + 1 + U[0,2^32), seed 3); every
    envelope must pass the restated verifier, and a sample must equal the oracle's bytes."""
    rng = np.random.default_rng(3)
    olds = rng.integers(0, 2**63, 1024, dtype=np.uint64)
    news = olds + 1 + rng.integers(0, 2**32, 1024, dtype=np.uint64)
    rc, proofs, st = run(hip, olds, news)
    assert rc == 0 and (st == 0).all()
    for i in range(0, 1024, 8):
        assert s.verify_improvement(proofs[i], int(olds[i])), i
    for i in (0, 511, 1023):
        assert proofs[i] == s.prove_improvement(int(olds[i]), int(news[i]))
    assert all(p[:2] == bytes([2, 5]) and len(p) <= 3527 for p in proofs)


def test_invalid_ops_are_reported_per_item(hip):
    rc, proofs, st = run(hip, [10, 7, 3], [20, 7, 2])
    assert rc == 1 and list(st) == [0, 1, 1] and proofs[1] == b"" and proofs[2] == b""
    assert proofs[0] == s.prove_improvement(10, 20)


def test_python_api(hip):
    import libzkp_amd as z
    assert z.prove_improvement(30, 50) == s.prove_improvement(30, 50)
    with pytest.raises(ValueError, match="new value must be greater than old value"):
        z.prove_improvement(50, 30)
    b = z.create_proof_batch()
    z.batch_add_improvement_proof(b, 1, 9)
    z.batch_add_range_proof(b, 5, 0, 10)
    z.batch_add_improvement_proof(b, 100, 250)
    assert z.get_batch_status(b)["improvement_proofs"] == 2
    out = z.process_batch(b, seeds=bytes(96))
    assert out[0] == s.prove_improvement(1, 9) and out[2] == s.prove_improvement(100, 250) and len(out[1]) == 1478
    res = z.benchmark_proof_generation_numeric("improvement", 3)
    assert res["successful_iterations"] == 3.0


def test_device_pointer_entry(hip):
    import torch
    n = 256
    rng = np.random.default_rng(11)
    olds = rng.integers(0, 2**62, n, dtype=np.uint64)
    news = olds + 1 + rng.integers(0, 2**20, n, dtype=np.uint64)
    stride = int(hip.zkp_hip_improvement_max_bytes())
    d_o = torch.from_numpy(olds.view(np.int64)).cuda()
    d_w = torch.from_numpy(news.view(np.int64)).cuda()
    d_out = torch.zeros((n, stride), dtype=torch.uint8, device="cuda")
    d_len = torch.zeros(n, dtype=torch.int32, device="cuda")
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        rc = hip.zkp_hip_prove_improvement_batch_device(n, ctypes.c_void_p(d_o.data_ptr()), ctypes.c_void_p(d_w.data_ptr()),
                                                        ctypes.c_void_p(d_out.data_ptr()), stride, ctypes.c_void_p(d_len.data_ptr()),
                                                        ctypes.c_void_p(stream.cuda_stream))
    assert rc == 0
    stream.synchronize()
    out, ln = d_out.cpu().numpy(), d_len.cpu().numpy()
    for i in (0, 100, 255):
        assert out[i, :ln[i]].tobytes() == s.prove_improvement(int(olds[i]), int(news[i]))


def test_verification_matches_oracle(hip):
    import libzkp_amd as z
    rng = np.random.default_rng(23)
    n = 200
    olds = rng.integers(0, 2**63, n, dtype=np.uint64)
    news = olds + 1 + rng.integers(0, 2**32, n, dtype=np.uint64)
    rc, proofs, st = run(hip, olds, news)
    assert rc == 0
    assert z.verify_improvement_batch(proofs, [int(x) for x in olds]) == [True] * n
    assert z.verify_improvement_batch(proofs, [int(x) + 1 for x in olds]) == [False] * n
    bad = []
    for p in proofs:
        b = bytearray(p); b[int(rng.integers(0, len(p)))] ^= 1 << int(rng.integers(0, 8)); bad.append(bytes(b))
    want = [s.verify_improvement(b, int(o)) for b, o in zip(bad, olds)]
    assert z.verify_improvement_batch(bad, [int(x) for x in olds]) == want and not any(want)
    assert not z.verify_improvement(b"", 0) and not z.verify_improvement(proofs[0][:-1], int(olds[0]))
    assert z.verify_improvement(s.prove_improvement(30, 50), 30)                  # an oracle-made proof
