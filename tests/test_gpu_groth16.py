"""Groth16/BN254 on the MI355X through the C ABI against the oracle: bit-exact vs the toxic-waste prover (an independent
route: no MSM, no FFT), pairing verification, the reference's accept/reject cases (snark.rs:617-641)."""
import ctypes
import json
import os

import numpy as np
import pytest

from oracle.py import groth16 as g
from util import P

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
SS = bytes(range(32))          # setup seed of the committed test keys (tests/golden/gen_groth16_keys.py)


@pytest.fixture(scope="module")
def hip():
    from libzkp_amd import _native
    L = _native.lib()
    _native.check(L.zkp_hip_init(0), "zkp_hip_init")
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        blob = open(os.path.join(GOLD, name), "rb").read()
        assert L.zkp_hip_groth16_load_key(kind, blob, len(blob)) == 0, _native.last_error()
    return L


def _rs(seed):
    return g.draw_fr(seed, 0x47313600, 0), g.draw_fr(seed, 0x47313600, 1)


def ref_equality(value, seed):
    cm = g.commit_value_snark(value)
    cs = g.equality_circuit(value, value, int.from_bytes(cm, "little"))
    return g.envelope(2, g.prove_with_trapdoor(g.equality_key(SS), cs, *_rs(seed)), cm)


def ref_membership(value, the_set, seed):
    cm = g.commit_value_snark(value)
    sel, sv, ir = g.membership_inputs(value, the_set)
    cs = g.membership_circuit(value, sel, sv, ir, int.from_bytes(cm, "little"))
    pr = g.prove_with_trapdoor(g.membership_key(SS), cs, *_rs(seed))
    return g.envelope(4, len(the_set).to_bytes(4, "little") + b"".join(x.to_bytes(8, "little") for x in the_set) + pr, cm)


def test_mimc_commitments(hip):
    vec = json.load(open(os.path.join(GOLD, "groth16_vectors.json")))["mimc"]
    vals = np.array([int(k) for k in vec], dtype=np.uint64)
    out = np.zeros((len(vals), 32), dtype=np.uint8)
    assert hip.zkp_hip_snark_commit_value_batch(len(vals), P(vals), P(out)) == 0
    assert [out[i].tobytes().hex() for i in range(len(vals))] == [vec[str(int(v))] for v in vals]
    rng = np.random.default_rng(3)
    vals = rng.integers(0, 2**63, 300, dtype=np.uint64)
    out = np.zeros((300, 32), dtype=np.uint8)
    assert hip.zkp_hip_snark_commit_value_batch(300, P(vals), P(out)) == 0
    assert all(out[i].tobytes() == g.commit_value_snark(int(vals[i])) for i in range(0, 300, 7))


def test_equality_bit_exact_and_verified(hip):
    rng = np.random.default_rng(11)
    n = 70
    v = rng.integers(0, 2**63, n, dtype=np.uint64)
    v[:4] = [42, 0, 2**64 - 1, 1]
    seeds = rng.integers(0, 256, 32 * n, dtype=np.uint8)
    out = np.zeros((n, 320), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint32)
    st = np.zeros(n, dtype=np.int32)
    assert hip.zkp_hip_prove_equality_batch(n, P(v), P(v), P(seeds), P(out), 320, P(lens), P(st)) == 0
    assert (lens == 298).all() and (st == 0).all() and (out[:, 298:] == 0).all()
    for i in list(range(12)) + [n - 1]:
        assert out[i, :298].tobytes() == ref_equality(int(v[i]), seeds[32 * i: 32 * i + 32].tobytes()), i
    env = out[0, :298].tobytes()
    cm = g.commit_value_snark(42)
    assert g.verify_equality_with_commitment(env, cm, SS)                                  # snark.rs:637
    wrong = g.commit_value_snark(99)
    assert not g.verify_equality_with_commitment(env[:266] + wrong, wrong, SS)             # snark.rs:639-640
    bad = bytearray(env)
    bad[12] ^= 1
    assert not g.verify_equality_with_commitment(bytes(bad), cm, SS)
    assert g.verify_equality_with_commitment(out[n - 1, :298].tobytes(), g.commit_value_snark(int(v[n - 1])), SS)


def test_equality_invalid_ops_and_randomness(hip):
    a = np.array([5, 6, 7], dtype=np.uint64)
    b = np.array([5, 60, 7], dtype=np.uint64)
    out = np.zeros((3, 298), dtype=np.uint8)
    lens = np.zeros(3, dtype=np.uint32)
    st = np.zeros(3, dtype=np.int32)
    assert hip.zkp_hip_prove_equality_batch(3, P(a), P(b), None, P(out), 298, P(lens), P(st)) == 1
    assert list(st) == [0, 1, 0] and list(lens) == [298, 0, 298] and (out[1] == 0).all()
    first = out.copy()
    assert hip.zkp_hip_prove_equality_batch(3, P(a), P(b), None, P(out), 298, P(lens), P(st)) == 1
    assert not (first[0] == out[0]).all()                                                  # fresh r, s each call (OsRng semantics)
    assert (first[0, 266:] == out[0, 266:]).all()                                          # commitment is deterministic
    assert g.verify_equality_with_commitment(out[2].tobytes(), g.commit_value_snark(7), SS)
    assert hip.zkp_hip_prove_equality_batch(3, P(a), P(b), None, P(out), 100, P(lens), P(st)) == -3


def test_membership_bit_exact_and_verified(hip):
    cases = [(25, [10, 20, 25, 30, 40]), (5, [5]), (9, [7, 9, 9]), (63, list(range(64))), (0, [3, 0]), (2**64 - 1, [1, 2**64 - 1]),
             (4, [1, 2, 3]), (1, [])]
    vals = np.array([c[0] for c in cases], dtype=np.uint64)
    flat = np.array([x for c in cases for x in c[1]] + [0], dtype=np.uint64)
    cnt = np.array([len(c[1]) for c in cases], dtype=np.uint32)
    n = len(cases)
    seeds = (np.arange(32 * n, dtype=np.uint32) * 5 + 2).astype(np.uint8)
    stride = 10 + 4 + 8 * 64 + 256 + 32
    out = np.zeros((n, stride), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint32)
    st = np.zeros(n, dtype=np.int32)
    assert hip.zkp_hip_prove_membership_batch(n, P(vals), P(flat), P(cnt), P(seeds), P(out), stride, P(lens), P(st)) == 1
    assert list(st) == [0, 0, 0, 0, 0, 0, 1, 1]
    for i in range(6):
        value, the_set = cases[i]
        assert lens[i] == 10 + 4 + 8 * len(the_set) + 256 + 32
        assert out[i, : lens[i]].tobytes() == ref_membership(value, the_set, seeds[32 * i: 32 * i + 32].tobytes()), i
    env = out[0, : lens[0]].tobytes()
    assert g.verify_membership(env, [10, 20, 25, 30, 40], SS)
    assert not g.verify_membership(env, [10, 20, 25, 30, 41], SS)                          # tests/integration.rs:87-91
    assert g.verify_membership(out[3, : lens[3]].tobytes(), list(range(64)), SS)


def test_native_key_generation_matches_oracle_setup(hip):
    """Trusted setup on the GPU from the same setup seed == oracle setup == the committed key files, byte for byte;
    a key from OS randomness differs and still proves."""
    from libzkp_amd import _native
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        pk, vk = ctypes.create_string_buffer(1 << 20), ctypes.create_string_buffer(1 << 16)
        pl, vl = ctypes.c_uint64(), ctypes.c_uint64()
        assert hip.zkp_hip_groth16_generate_key(kind, SS, pk, 1 << 20, ctypes.byref(pl), vk, 1 << 16, ctypes.byref(vl)) == 0, _native.last_error()
        ref = open(os.path.join(GOLD, name), "rb").read()
        assert pk.raw[: pl.value] == ref
        assert vk.raw[: vl.value] == ref[: vl.value] and vl.value == 64 + 3 * 128 + 8 + 64 * (2 if kind == 0 else 130)
    pl = ctypes.c_uint64()
    pk = ctypes.create_string_buffer(1 << 20)
    assert hip.zkp_hip_groth16_generate_key(0, None, pk, 1 << 20, ctypes.byref(pl), None, 0, None) == 0
    assert pk.raw[: pl.value] != open(os.path.join(GOLD, "equality_mimc_pk.bin"), "rb").read()
    a = np.array([77], dtype=np.uint64)
    out = np.zeros((1, 298), dtype=np.uint8)
    lens = np.zeros(1, dtype=np.uint32)
    st = np.zeros(1, dtype=np.int32)
    assert hip.zkp_hip_prove_equality_batch(1, P(a), P(a), None, P(out), 298, P(lens), P(st)) == 0
    assert not g.verify_equality_with_commitment(out[0].tobytes(), g.commit_value_snark(77), SS)      # proof under the NEW key
    blob = open(os.path.join(GOLD, "equality_mimc_pk.bin"), "rb").read()                                # restore the test key
    assert hip.zkp_hip_groth16_load_key(0, blob, len(blob)) == 0
    assert hip.zkp_hip_groth16_load_key(0, blob[:-1], len(blob) - 1) == -3                              # truncated key rejected


def test_python_api_generates_and_persists_keys(hip, tmp_path):
    import libzkp_amd as z
    from libzkp_amd import api
    api._keys_loaded.clear()
    api._key_dir_override = None
    z.set_snark_key_dir(str(tmp_path))
    p = z.prove_equality(9, 9)
    assert len(p) == 298 and sorted(os.listdir(tmp_path)) == ["equality_mimc_pk.bin", "equality_mimc_vk.bin"]
    pk = open(tmp_path / "equality_mimc_pk.bin", "rb").read()
    assert len(pk) == os.path.getsize(os.path.join(GOLD, "equality_mimc_pk.bin"))
    api._keys_loaded.clear()                                                   # a second "process" loads the persisted key
    q = z.prove_equality(9, 9)
    assert q[266:] == p[266:] and q != p


def test_python_api_snark(hip, monkeypatch):
    import libzkp_amd as z
    from libzkp_amd import api
    api._keys_loaded.clear()
    api._key_dir_override = None
    z.set_snark_key_dir(GOLD)
    assert not z.is_snark_setup_initialized()
    assert z.snark_commit_value(42) == g.commit_value_snark(42)
    p = z.prove_equality(42, 42)
    assert z.is_snark_setup_initialized() and len(p) == 298
    assert g.verify_equality_with_commitment(p, z.snark_commit_value(42), SS)
    with pytest.raises(TypeError, match="already initialized"):
        z.set_snark_key_dir("/elsewhere")
    m = z.prove_membership(25, [10, 20, 25, 30, 40])
    assert g.verify_membership(m, [10, 20, 25, 30, 40], SS)
    b = z.create_proof_batch()                                                             # examples/demo.rs:66-105 shape
    z.batch_add_range_proof(b, 25, 18, 65)
    z.batch_add_equality_proof(b, 100, 100)
    z.batch_add_threshold_proof(b, [100, 200, 300], 500)
    z.batch_add_membership_proof(b, 7, [3, 7, 9])
    z.batch_add_consistency_proof(b, [10, 20, 30])
    z.batch_add_equality_proof(b, 5, 5)
    proofs = z.process_batch(b)
    assert [p[1] for p in proofs] == [1, 2, 3, 4, 6, 2]
    assert g.verify_equality_with_commitment(proofs[1], g.commit_value_snark(100), SS)
    assert g.verify_membership(proofs[3], [3, 7, 9], SS)
    r = z.benchmark_proof_generation("equality", 2)
    assert r["proof_type"] == "equality" and float(r["success_rate"]) == 100.0


def _prove_some(L):
    """64 equality + 32 membership proofs with fixed seeds through the per-variant entry points; returns the bytes."""
    from libzkp_amd import workloads as wl
    from util import P
    n = 64
    rng = np.random.default_rng(77)
    a = rng.integers(0, 2**64, n, dtype=np.uint64)
    sd = wl.op_seeds(78, n)
    o = np.zeros((n, 298), dtype=np.uint8); ln = np.zeros(n, dtype=np.uint32); st = np.zeros(n, dtype=np.int32)
    assert L.zkp_hip_prove_equality_batch(n, P(a), P(a), P(sd), P(o), 298, P(ln), P(st)) == 0 and not st.any()
    m = 32
    sets = np.stack([rng.choice(2**32, 16, replace=False) for _ in range(m)]).astype(np.uint64)
    vals = sets[np.arange(m), np.arange(m) % 16].copy()
    cnt = np.full(m, 16, dtype=np.uint32)
    stride = 10 + 4 + 8 * 16 + 256 + 32
    o2 = np.zeros((m, stride), dtype=np.uint8); l2 = np.zeros(m, dtype=np.uint32); s2 = np.zeros(m, dtype=np.int32)
    flat = np.ascontiguousarray(sets.ravel())
    assert L.zkp_hip_prove_membership_batch(m, P(vals), P(flat), P(cnt), P(sd[:32 * m].copy()), P(o2), stride, P(l2), P(s2)) == 0 and not s2.any()
    return o.tobytes() + o2.tobytes()


def test_key_tables_are_sized_at_load_time_and_the_bytes_do_not_change():
    """The radix of a key's window tables is chosen when the key is loaded, from the memory budget (snark.rs:40-70,122-139 loads a 140 KB
    key into any process): with ~1.2 GB per key the equality key falls back from radix 2^14 to 2^9 and the membership key to 2^8, the
    proofs stay bit-identical; with a budget nothing fits into, the load fails with a message that says so."""
    from libzkp_amd import _native
    L = _native.lib()
    _native.check(L.zkp_hip_init(0), "zkp_hip_init")
    keys = [(kind, open(os.path.join(GOLD, name), "rb").read()) for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin"))]
    def load():
        return [L.zkp_hip_groth16_load_key(kind, blob, len(blob)) for kind, blob in keys]
    assert load() == [0, 0], _native.last_error()
    want = _prove_some(L)
    try:
        os.environ["ZKP_HIP_G16_TABLE_BUDGET_MB"] = "1200"
        assert load() == [0, 0], _native.last_error()
        assert _prove_some(L) == want
        for wb in ("11", "13"):
            os.environ["ZKP_HIP_G16_WBITS"] = wb
            assert load() == [0, 0], _native.last_error()
            assert _prove_some(L) == want, wb
        del os.environ["ZKP_HIP_G16_WBITS"]
        os.environ.pop("ZKP_HIP_G16_TABLE_BUDGET_MB", None)
        os.environ["ZKP_HIP_G16_UNEVEN"] = "0"                     # radix 2^14 in its even form (19 windows) against the default's uneven form (18)
        assert load() == [0, 0], _native.last_error()
        assert _prove_some(L) == want, "even 2^14"
        del os.environ["ZKP_HIP_G16_UNEVEN"]
        os.environ["ZKP_HIP_G16_TABLE_BUDGET_MB"] = "100"
        assert load()[0] < 0 and "not enough device memory" in _native.last_error()
    finally:
        os.environ.pop("ZKP_HIP_G16_TABLE_BUDGET_MB", None); os.environ.pop("ZKP_HIP_G16_WBITS", None); os.environ.pop("ZKP_HIP_G16_UNEVEN", None)
        assert load() == [0, 0], _native.last_error()
    assert _prove_some(L) == want


def test_shards_of_one_gpu_share_a_key_s_tables():
    import torch
    from libzkp_amd import _native
    L = _native.lib()
    keys = [(kind, open(os.path.join(GOLD, name), "rb").read()) for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin"))]
    L.zkp_hip_shutdown()
    try:
        _native.init_devices([0])
        for kind, blob in keys:
            assert L.zkp_hip_groth16_load_key(kind, blob, len(blob)) == 0, _native.last_error()
        free1 = torch.cuda.mem_get_info(0)[0]
        L.zkp_hip_shutdown()
        _native.init_devices([0, 0])
        for kind, blob in keys:
            assert L.zkp_hip_groth16_load_key(kind, blob, len(blob)) == 0, _native.last_error()
        free2 = torch.cuda.mem_get_info(0)[0]
        assert free1 - free2 < 8 << 30, (free1 >> 20, free2 >> 20)          # a second copy of the tables would be tens of GB
        assert _prove_some(L)                                               # (shard 0 of the two)
    finally:
        L.zkp_hip_shutdown()
        _native.check(L.zkp_hip_init(0), "zkp_hip_init")
        for kind, blob in keys:
            assert L.zkp_hip_groth16_load_key(kind, blob, len(blob)) == 0
