"""Batch sizes around the kernels' granularities (64-lane waves, 256/512/1024-lane MSM workgroups, chunk rounds): every
proof of an oddly sized batch must still be right.  Range: GPU verifier over all + oracle bytes on a sample; Groth16:
bit-exact trapdoor reference on a sample; STARK: oracle bytes on a sample and envelope sanity on all."""
import hashlib
import os

import numpy as np
import pytest

from oracle.py import groth16 as g
from oracle.py import stark
from util import P, oracle_prove, workload

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SS = bytes(range(32))


@pytest.fixture(scope="module")
def hip():
    from libzkp_amd import _native
    L = _native.lib()
    _native.check(L.zkp_hip_init(0), "zkp_hip_init")
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        blob = open(os.path.join(GOLD, name), "rb").read()
        assert L.zkp_hip_groth16_load_key(kind, blob, len(blob)) == 0
    return L


@pytest.mark.parametrize("n", [1, 2, 3, 31, 33, 511, 513, 1025, 2049, 5000])
def test_range_sizes(hip, oracle_c, n):
    v, mn, mx, seeds = workload(n, 100 + n)
    out = np.zeros((n, 1478), dtype=np.uint8); lens = np.zeros(n, dtype=np.uint32); st = np.zeros(n, dtype=np.int32)
    assert hip.zkp_hip_prove_range_batch(n, P(v), P(mn), P(mx), 64, P(seeds), P(out), 1478, P(lens), P(st)) == 0
    ok = np.zeros(n, dtype=np.uint8)
    assert hip.zkp_hip_verify_range_batch(n, P(out), 1478, P(lens), P(mn), P(mx), P(ok)) == 0 and (ok == 1).all()
    idx = sorted({0, n - 1, n // 2, (7 * n) // 13})
    rc, ref, _, _ = oracle_prove(oracle_c, v[idx], mn[idx], mx[idx], np.concatenate([seeds[32 * i:32 * i + 32] for i in idx]))
    assert rc == 0 and all(out[i].tobytes() == ref[k].tobytes() for k, i in enumerate(idx))


@pytest.mark.parametrize("n", [1, 5, 257, 1023, 1025, 3000])
def test_equality_sizes(hip, n):
    rng = np.random.default_rng(200 + n)
    v = rng.integers(0, 2**63, n, dtype=np.uint64)
    seeds = rng.integers(0, 256, 32 * n, dtype=np.uint8)
    out = np.zeros((n, 298), dtype=np.uint8); lens = np.zeros(n, dtype=np.uint32); st = np.zeros(n, dtype=np.int32)
    assert hip.zkp_hip_prove_equality_batch(n, P(v), P(v), P(seeds), P(out), 298, P(lens), P(st)) == 0 and (lens == 298).all()
    for i in sorted({0, n - 1, n // 2}):
        sd = seeds[32 * i:32 * i + 32].tobytes()
        cm = g.commit_value_snark(int(v[i]))
        cs = g.equality_circuit(int(v[i]), int(v[i]), int.from_bytes(cm, "little"))
        want = g.envelope(2, g.prove_with_trapdoor(g.equality_key(SS), cs, g.draw_fr(sd, 0x47313600, 0), g.draw_fr(sd, 0x47313600, 1)), cm)
        assert out[i].tobytes() == want, (n, i)
    # the GPU verifier over the whole oddly sized batch: all accepted; a flipped bit in A, B, C or the commitment of every third rejected
    ok = np.zeros(n, dtype=np.uint8)
    assert hip.zkp_hip_verify_equality_batch(n, P(out), 298, P(lens), P(ok)) == 0 and (ok == 1).all()
    t = out.copy()
    pos = np.array([12, 80, 210, 270])[np.arange(n) % 4]
    t[np.arange(0, n, 3), pos[::3]] ^= 4
    assert hip.zkp_hip_verify_equality_batch(n, P(t), 298, P(lens), P(ok)) == 0
    want_ok = np.ones(n, dtype=np.uint8); want_ok[::3] = 0
    assert (ok == want_ok).all()


@pytest.mark.parametrize("n", [1, 4, 300, 1025])
def test_membership_sizes(hip, n):
    rng = np.random.default_rng(300 + n)
    sets = rng.integers(0, 2**32, (n, 5), dtype=np.uint64)
    v = sets[np.arange(n), np.arange(n) % 5].copy()
    cnt = np.full(n, 5, dtype=np.uint32); flat = sets.ravel().copy()
    seeds = rng.integers(0, 256, 32 * n, dtype=np.uint8)
    stride = 10 + 4 + 8 * 5 + 256 + 32
    out = np.zeros((n, stride), dtype=np.uint8); lens = np.zeros(n, dtype=np.uint32); st = np.zeros(n, dtype=np.int32)
    assert hip.zkp_hip_prove_membership_batch(n, P(v), P(flat), P(cnt), P(seeds), P(out), stride, P(lens), P(st)) == 0 and (lens == stride).all()
    for i in sorted({0, n - 1}):
        assert g.verify_membership(out[i].tobytes(), [int(x) for x in sets[i]], SS), (n, i)


@pytest.mark.parametrize("n", [1, 63, 65, 1000, 70000])
def test_improvement_sizes(hip, n):
    rng = np.random.default_rng(400 + n)
    olds = rng.integers(0, 2**63, n, dtype=np.uint64)
    news = olds + 1 + rng.integers(0, 2**32, n, dtype=np.uint64)
    stride = int(hip.zkp_hip_improvement_max_bytes())
    out = np.zeros((n, stride), dtype=np.uint8); lens = np.zeros(n, dtype=np.uint32); st = np.zeros(n, dtype=np.int32)
    assert hip.zkp_hip_prove_improvement_batch(n, P(olds), P(news), P(out), stride, P(lens), P(st)) == 0 and (st == 0).all()
    assert (out[:, 0] == 2).all() and (out[:, 1] == 5).all() and (lens >= 1200).all() and (lens <= stride).all()
    for i in sorted({0, n - 1, n // 3, int(lens.argmin()), int(lens.argmax())}):      # incl. the shortest and longest proof of the batch
        assert out[i, :lens[i]].tobytes() == stark.prove_improvement(int(olds[i]), int(news[i]))
    # the binding commitment (last 32 bytes) of every envelope
    for i in range(0, n, max(1, n // 50)):
        want = hashlib.sha256(b"libzkp_improvement_v1" + int(olds[i]).to_bytes(8, "little") + int(news[i]).to_bytes(8, "little")).digest()
        assert out[i, lens[i] - 32:lens[i]].tobytes() == want


def test_concurrent_callers(hip, oracle_c):
    """The reference calls its backends from rayon workers (batch.rs:125-130): concurrent calls into the C ABI from
    several host threads must each get their own, correct results (the library serialises device work internally)."""
    import threading
    results, errors = {}, []

    def worker(t):
        try:
            n = 40 + 7 * t
            v, mn, mx, seeds = workload(n, 900 + t)
            out = np.zeros((n, 1478), dtype=np.uint8); lens = np.zeros(n, dtype=np.uint32); st = np.zeros(n, dtype=np.int32)
            for _ in range(3):
                assert hip.zkp_hip_prove_range_batch(n, P(v), P(mn), P(mx), 64, P(seeds), P(out), 1478, P(lens), P(st)) == 0
                olds = np.arange(1, n + 1, dtype=np.uint64) * (t + 1); news = olds + 5
                so = np.zeros((n, 3527), dtype=np.uint8); sl = np.zeros(n, dtype=np.uint32); ss = np.zeros(n, dtype=np.int32)
                assert hip.zkp_hip_prove_improvement_batch(n, P(olds), P(news), P(so), 3527, P(sl), P(ss)) == 0
            results[t] = (v, mn, mx, seeds, out.copy(), so[0, :sl[0]].tobytes(), int(olds[0]), int(news[0]))
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    for t, (v, mn, mx, seeds, out, sp, o, w) in results.items():
        rc, ref, _, _ = oracle_prove(oracle_c, v, mn, mx, seeds)
        assert rc == 0 and (out == ref).all(), t
        assert sp == stark.prove_improvement(o, w)
