"""Batched range-proof verification on the MI355X through the C ABI against the oracle's verifier: identical accept /
reject verdicts on proofs made by the GPU prover, on tampered envelopes, wrong bounds and malformed input."""
import numpy as np
import pytest

from util import P, oracle_prove, oracle_verify, workload

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from libzkp_amd import _native
    L = _native.lib()
    _native.check(L.zkp_hip_init(0), "zkp_hip_init")
    return L


def gpu_verify(L, out, lens, mn, mx):
    n = out.shape[0]
    ok = np.zeros(n, dtype=np.uint8)
    assert L.zkp_hip_verify_range_batch(n, P(out), out.shape[1], P(lens), P(mn), P(mx), P(ok)) == 0
    return ok


def gpu_prove(L, v, mn, mx, seeds, stride=1478):
    n = len(v)
    out = np.zeros((n, stride), dtype=np.uint8); lens = np.zeros(n, dtype=np.uint32); st = np.zeros(n, dtype=np.int32)
    assert L.zkp_hip_prove_range_batch(n, P(v), P(mn), P(mx), 64, P(seeds), P(out), stride, P(lens), P(st)) == 0
    return out, lens


def test_gpu_proofs_verify_and_tampering_matches_oracle(hip, oracle_c):
    rng = np.random.default_rng(41)
    n = 300
    v, mn, mx, seeds = workload(n, 5)
    v[0], v[1] = 0, 2**32
    out, lens = gpu_prove(hip, v, mn, mx, seeds)
    assert (gpu_verify(hip, out, lens, mn, mx) == 1).all()
    # one flipped bit per envelope at a random position: verdicts must equal the oracle's
    t = out.copy()
    pos = rng.integers(0, 1478, n)
    t[np.arange(n), pos] ^= (1 << rng.integers(0, 8, n)).astype(np.uint8)
    want = oracle_verify(oracle_c, t, lens, mn, mx)[1]
    got = gpu_verify(hip, t, lens, mn, mx)
    assert (got == want).all() and want.sum() == 0
    # a batch mixing valid and invalid envelopes keeps them apart
    mix = out.copy(); mix[1::2] = t[1::2]
    got = gpu_verify(hip, mix, lens, mn, mx)
    assert (got[0::2] == 1).all() and (got[1::2] == 0).all()
    # wrong bounds (bulletproofs.rs:704), truncated length, min > max
    assert (gpu_verify(hip, out, lens, mn + 1, mx) == 0).all()
    assert (gpu_verify(hip, out, lens, mn, mx - 1) == 0).all()
    assert (gpu_verify(hip, out, lens - 1, mn, mx) == 0).all()
    assert (gpu_verify(hip, out, lens, mx, mn) == 0).all()


def test_narrow_bounds_and_stride_padding(hip, oracle_c):
    n = 40
    v, mn, mx, seeds = workload(n, 9, lo=1000, hi=1000 + 2**20)
    out, lens = gpu_prove(hip, v, mn, mx, seeds, stride=1600)           # padded records
    assert (gpu_verify(hip, out, lens, mn, mx) == 1).all()
    assert (oracle_verify(oracle_c, out, lens, mn, mx)[1] == 1).all()
    garbage = np.random.default_rng(1).integers(0, 256, out.shape, dtype=np.uint8)
    assert (gpu_verify(hip, garbage, lens, mn, mx) == 0).all()
    zero_len = np.zeros(n, dtype=np.uint32)
    assert (gpu_verify(hip, out, zero_len, mn, mx) == 0).all()


def test_oracle_made_proofs_and_python_api(hip, oracle_c):
    import libzkp_amd as z
    v, mn, mx, seeds = workload(8, 3)
    rc, out, lens, st = oracle_prove(oracle_c, v, mn, mx, seeds)
    assert rc == 0 and (gpu_verify(hip, out, lens, mn, mx) == 1).all()
    proofs = [out[i].tobytes() for i in range(8)]
    assert z.verify_range_batch(proofs, [0] * 8, [2**32] * 8) == [True] * 8
    assert z.verify_range(proofs[0], 0, 2**32) and not z.verify_range(proofs[0], 1, 2**32)
    assert not z.verify_range(proofs[0][:-1], 0, 2**32) and not z.verify_range(b"", 0, 1) and not z.verify_range(b"\x02\x01" + bytes(5000), 0, 1)
    p = z.prove_range(50, 0, 100)
    assert z.verify_range(p, 0, 100) and not z.verify_range(p, 0, 99)


def test_full_batch_of_4096(hip):
    v, mn, mx, seeds = workload(4096, 1)
    out, lens = gpu_prove(hip, v, mn, mx, seeds)
    import time
    ok = gpu_verify(hip, out, lens, mn, mx)
    t0 = time.perf_counter(); ok = gpu_verify(hip, out, lens, mn, mx); dt = time.perf_counter() - t0
    assert (ok == 1).all()
    print("verified 4096 range envelopes in %.2f ms (host buffers)" % (dt * 1e3))


def test_batch_check_and_per_job_check_agree(hip, oracle_c, monkeypatch):
    """The random-linear-combination batch check (bpv_impl.inc, default from 4096 jobs up) forced on at 300 envelopes and forced
    off: the same verdicts as the oracle either way -- a clean batch accepted in one piece, any bad envelope sends the batch
    through the per-job check, which names exactly the bad ones."""
    n = 300
    v, mn, mx, seeds = workload(n, 11)
    out, lens = gpu_prove(hip, v, mn, mx, seeds)
    rng = np.random.default_rng(5)
    bad = out.copy()
    rows = rng.choice(n, 7, replace=False)
    bad[rows, rng.integers(2, 1478, 7)] ^= 0x10
    want = oracle_verify(oracle_c, bad, lens, mn, mx)[1]
    assert want.sum() == n - 7
    for env in ({"ZKP_HIP_BATCH_VERIFY_MIN": "64"}, {"ZKP_HIP_NO_BATCH_VERIFY": "1"}):
        for k, val in env.items():
            monkeypatch.setenv(k, val)
        assert (gpu_verify(hip, out, lens, mn, mx) == 1).all()
        assert (gpu_verify(hip, bad, lens, mn, mx) == want).all()
        assert (gpu_verify(hip, out, lens, mn + 1, mx) == 0).all()
        one = out.copy(); one[n - 1, 40] ^= 1                        # a single bad envelope at the end of the batch
        got = gpu_verify(hip, one, lens, mn, mx)
        assert got[:-1].all() and got[-1] == 0
        for k in env:
            monkeypatch.delenv(k)


@pytest.fixture(params=["default", "batch-check-first"])
def verify_mode(request, monkeypatch):
    """the Bulletproofs verifiers as they run by default at these sizes (per-job check) and with the whole-batch check forced on"""
    if request.param == "batch-check-first":
        monkeypatch.setenv("ZKP_HIP_BATCH_VERIFY_MIN", "1")
    return request.param


def test_threshold_verification_matches_oracle(hip, oracle_c, verify_mode):
    import ctypes
    import libzkp_amd as z
    rng = np.random.default_rng(17)
    n = 64
    lists = [[int(x) for x in rng.integers(0, 2**40, int(k))] for k in rng.integers(1, 6, n)]
    ths = [int(rng.integers(0, sum(v) + 1)) for v in lists]
    ths[0] = sum(lists[0])                                     # tight: sum == threshold
    proofs = z.prove_threshold_batch(lists, ths, seeds=bytes(rng.integers(0, 256, 32 * n, dtype=np.uint8)))
    assert all(len(p) == 762 for p in proofs)
    assert z.verify_threshold_batch(proofs, ths) == [True] * n
    assert z.verify_threshold_batch(proofs, [t + 1 for t in ths]) == [False] * n          # wrong threshold
    # tampering: verdicts equal the oracle's
    orc = oracle_c
    orc.zkp_oracle_verify_threshold.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint64]
    bad = []
    for i, p in enumerate(proofs):
        b = bytearray(p); b[int(rng.integers(0, 762))] ^= 1 << int(rng.integers(0, 8)); bad.append(bytes(b))
    want = [bool(orc.zkp_oracle_verify_threshold(b, len(b), ths[i])) for i, b in enumerate(bad)]
    assert z.verify_threshold_batch(bad, ths) == want and not any(want)
    assert [bool(orc.zkp_oracle_verify_threshold(p, len(p), ths[i])) for i, p in enumerate(proofs[:8])] == [True] * 8
    assert z.verify_threshold(proofs[3], ths[3]) and not z.verify_threshold(proofs[3][:-1], ths[3]) and not z.verify_threshold(b"", 0)
    # a range envelope is not a threshold envelope
    assert not z.verify_threshold(z.prove_range(5, 0, 10), 0)


def test_consistency_verification_matches_oracle(hip, oracle_c, verify_mode):
    import ctypes
    import libzkp_amd as z
    rng = np.random.default_rng(29)
    data = []
    for k in (1, 2, 2, 3, 5, 8, 1, 4):
        data.append(sorted(int(x) for x in rng.integers(0, 2**50, k)))
    data[2] = [7, 7]                                           # equal neighbours: difference 0
    proofs = z.prove_consistency_batch(data, seeds=bytes(rng.integers(0, 256, 32 * len(data), dtype=np.uint8)))
    orc = oracle_c
    orc.zkp_oracle_verify_consistency.argtypes = [ctypes.c_char_p, ctypes.c_uint32]
    assert all(orc.zkp_oracle_verify_consistency(p, len(p)) == 1 for p in proofs)
    assert z.verify_consistency_batch(proofs) == [True] * len(proofs)
    # tampering anywhere (header, count, commitments, proofs, difference commitments, digest): verdicts equal the oracle's
    bad = []
    for p in proofs:
        for frac in (0.0, 0.01, 0.2, 0.5, 0.8, 0.97, 0.999):
            b = bytearray(p); i = min(len(p) - 1, int(frac * len(p))); b[i] ^= 1 << int(rng.integers(0, 8)); bad.append(bytes(b))
    want = [bool(orc.zkp_oracle_verify_consistency(b, len(b))) for b in bad]
    assert z.verify_consistency_batch(bad) == want and not any(want)
    assert not z.verify_consistency(b"") and not z.verify_consistency(proofs[4][:-1]) and not z.verify_consistency(proofs[0] + b"\\0")
    # a proof for a different list does not transplant
    sw = bytearray(proofs[3]); sw[14:46] = proofs[4][14:46]
    assert z.verify_consistency(bytes(sw)) == bool(orc.zkp_oracle_verify_consistency(bytes(sw), len(sw))) is False


def test_composite_and_parallel_verification(hip):
    """verify_composite_proof / verify_proofs_parallel (advanced/composite.rs:25-35, performance.rs:251-293) over the GPU verifiers"""
    import libzkp_amd as z
    r = z.prove_range(25, 18, 65)
    t = z.prove_threshold([10, 20, 30], 50)
    i = z.prove_improvement(30, 50)
    k = z.prove_consistency([1, 5, 9])
    comp = z.create_composite_proof([r, t, i, k])
    assert z.verify_composite_proof_integrity_only(comp) and z.verify_composite_proof(comp)
    meta = z.create_proof_with_metadata(r, {"purpose": b"age check"})
    assert z.verify_composite_proof(meta) and z.extract_proof_metadata(meta) == {"purpose": b"age check"}
    # an inner proof that is well-formed but cryptographically wrong: integrity passes, full verification does not
    bad_r = bytearray(r); bad_r[700] ^= 1
    comp_bad = z.create_composite_proof([bytes(bad_r), t])
    assert z.verify_composite_proof_integrity_only(comp_bad) and not z.verify_composite_proof(comp_bad)
    got = z.verify_proofs_parallel([(r, "range"), (t, "threshold"), (i, "improvement"), (k, "consistency"),
                                    (r, "threshold"), (bytes(bad_r), "range"), (b"junk", "range"), (i, "nope")])
    assert got == [True, True, True, True, False, False, False, False]


def test_groth16_verification(hip):
    """verify_equality / verify_membership (pairing check on the GPU) against the oracle's verifier; keys = the committed test keys"""
    import os
    import libzkp_amd as z
    import libzkp_amd.api as api
    from libzkp_amd import _native
    from oracle.py import groth16 as g
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        api.install_proving_key(kind, open(os.path.join(gold, name), "rb").read())
    SS = bytes(range(32))
    rng = np.random.default_rng(31)
    vals = [int(x) for x in rng.integers(0, 2**63, 24, dtype=np.uint64)]
    seeds = bytes(rng.integers(0, 256, 32 * 24, dtype=np.uint8))
    proofs = z.prove_equality_batch(vals, vals, seeds=seeds)
    assert all(z.verify_equality(p, v, v) for p, v in zip(proofs[:4], vals[:4]))
    assert z.verify_equality_with_commitment_batch(proofs, [z.snark_commit_value(v) for v in vals]) == [True] * 24
    assert not z.verify_equality(proofs[0], vals[0], vals[0] + 1) and not z.verify_equality(proofs[0], vals[1], vals[1])
    bad = []
    sign_bits = {(73, 7), (201, 7), (265, 7)}                # the sign flags of A, B, C: ignored for uncompressed points (below)
    for p in proofs:
        while True:
            pos, bit = int(rng.integers(0, 298)), int(rng.integers(0, 8))
            if (pos, bit) not in sign_bits:
                break
        b = bytearray(p); b[pos] ^= 1 << bit; bad.append(bytes(b))
    want = [g.verify_equality_with_commitment(b, b[266:], SS) for b in bad[:8]]
    got = api._verify_snark_envelopes(0, bad)
    assert got[:8] == want and not any(got)
    # ark-serialize's parsing rules on malleated encodings (snark.rs:378 -> Proof::deserialize_uncompressed): the sign flag of an
    # uncompressed finite point is ignored, so flipping it leaves the proof valid; both flag bits set is a deserialisation error
    mall = []
    for last in (73, 201, 265):
        b = bytearray(proofs[0]); b[last] ^= 0x80; mall.append(bytes(b))
    for last in (73, 201, 265):
        b = bytearray(proofs[0]); b[last] |= 0xC0; mall.append(bytes(b))
    want = [g.verify_equality_with_commitment(b, b[266:], SS) for b in mall]
    assert want == [True, True, True, False, False, False]
    assert api._verify_snark_envelopes(0, mall) == want
    sets = [[int(x) for x in rng.choice(2**32, 5, replace=False)] for _ in range(6)]
    mp = z.prove_membership_batch([s[i % 5] for i, s in enumerate(sets)], sets, seeds=seeds[:32 * 6])
    assert z.verify_membership_batch(mp, sets) == [True] * 6
    assert z.verify_membership(mp[0], list(reversed(sets[0]))) and not z.verify_membership(mp[0], sets[1])
    assert g.verify_membership(mp[2], sets[2], SS)
    b = bytearray(mp[3]); b[40] ^= 4
    assert not z.verify_membership(bytes(b), sets[3])
    # the composite / parallel front ends now cover all six schemes
    r = z.prove_range(25, 18, 65)
    comp = z.create_composite_proof([r, proofs[0], mp[0]])
    assert z.verify_composite_proof(comp)
    assert z.verify_proofs_parallel([(proofs[1], "equality"), (mp[1], "membership"), (bad[1], "equality"), (proofs[2], "membership")]) == [True, True, False, False]


def _f2_sqrt(bn, a):
    """square root in Fq2 (p = 3 mod 4), None for a non-residue"""
    def f2pow(b, e):
        r = (1, 0)
        while e:
            if e & 1:
                r = bn.f2_mul(r, b)
            b = bn.f2_mul(b, b); e >>= 1
        return r
    if a == (0, 0):
        return (0, 0)
    a1 = f2pow(a, (bn.P - 3) // 4)
    alpha = bn.f2_mul(bn.f2_mul(a1, a1), a)
    a0 = bn.f2_mul(f2pow(alpha, bn.P), alpha)
    if a0 == (bn.P - 1, 0):
        return None
    x0 = bn.f2_mul(a1, a)
    if alpha == (bn.P - 1, 0):
        return bn.f2_mul((0, 1), x0)
    return bn.f2_mul(f2pow(bn.f2_add((1, 0), alpha), (bn.P - 1) // 2), x0)


_G16_PATHS_CHILD = r"""
import json, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import libzkp_amd as z
import libzkp_amd.api as api
gold = os.path.join(sys.argv[1], "tests", "golden")
for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
    api.install_proving_key(kind, open(os.path.join(gold, name), "rb").read())
blobs = [bytes.fromhex(h) for h in json.load(open(sys.argv[2]))]
mblobs = [bytes.fromhex(h) for h in json.load(open(sys.argv[3]))]
print(json.dumps([api._verify_snark_envelopes(0, blobs[:-8]), api._verify_snark_envelopes(0, blobs), api._verify_snark_envelopes(1, mblobs)]))
"""


def test_groth16_machine_and_lane_per_chain_paths_agree(hip, tmp_path):
    """The Fq2 machine (fq2vm.h: the default) and the lane-per-chain kernels (ZKP_HIP_G16_VERIFY_VM=0, a fresh process) give the same
    verdicts on valid, bit-flipped and malleated envelopes; a batch that holds proofs with a point at infinity (valid encodings that
    the machine leaves to the other kernels) still gets every verdict right; a sample is checked against the oracle's verifier."""
    import json, os, subprocess, sys
    import libzkp_amd as z
    import libzkp_amd.api as api
    from oracle.py import groth16 as g
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        api.install_proving_key(kind, open(os.path.join(gold, name), "rb").read())
    SS = bytes(range(32))
    rng = np.random.default_rng(77)
    vals = [int(x) for x in rng.integers(0, 2**63, 150, dtype=np.uint64)]
    proofs = z.prove_equality_batch(vals, vals)
    blobs = list(proofs[:70])
    for p in proofs[70:150]:                                   # one to three flipped bits anywhere in the envelope
        b = bytearray(p)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, 298))] ^= 1 << int(rng.integers(0, 8))
        blobs.append(bytes(b))
    for last in (73, 201, 265):
        b = bytearray(proofs[0]); b[last] ^= 0x80; blobs.append(bytes(b))
    inf1 = bytes(63) + b"\x40"
    special = []
    for off, size in ((10, 64), (74, 128), (202, 64)):         # A, B, C replaced by the point at infinity
        b = bytearray(proofs[1]); b[off:off + size] = (bytes(size - 1) + b"\x40"); special.append(bytes(b))
    special += [bytes(bytearray(proofs[2][:10]) + inf1 + bytearray(proofs[2][74:]))] * 5
    # B moved out of G2 by a point of the twist's cofactor part (on the curve, canonical, finite): only the subgroup check rejects it
    from oracle.py import bn254 as bn
    import random as _random
    rr = _random.Random(9)
    while True:
        x = (rr.randrange(bn.P), rr.randrange(bn.P))
        y2 = bn.f2_add(bn.f2_mul(bn.f2_sq(x), x), bn.B2)
        y = _f2_sqrt(bn, y2)
        if y is not None:
            break
    cof = bn.G2C.mul_pt((x, y), bn.R, reduce=False)
    okb, bpt = bn.de_g2(proofs[3][74:202])
    assert okb and cof is not None and bn.G2C.is_on_curve(cof)
    moved = bytearray(proofs[3]); moved[74:202] = bn.ser_g2(bn.G2C.add_pts(bpt, cof))
    blobs.insert(153, bytes(moved))                             # index 153: checked against the oracle below
    blobs += special                                            # the last eight: present only in the second call of the child
    sets = [[int(x) for x in rng.choice(2**32, 7, replace=False)] for _ in range(20)]
    mp = z.prove_membership_batch([s[i % 7] for i, s in enumerate(sets)], sets)
    mblobs = list(mp[:10])
    for p in mp[10:]:
        b = bytearray(p); b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8)); mblobs.append(bytes(b))
    f1, f2 = tmp_path / "eq.json", tmp_path / "mem.json"
    f1.write_text(json.dumps([b.hex() for b in blobs])); f2.write_text(json.dumps([b.hex() for b in mblobs]))
    got = [api._verify_snark_envelopes(0, blobs[:-8]), api._verify_snark_envelopes(0, blobs), api._verify_snark_envelopes(1, mblobs)]
    env = dict(os.environ, ZKP_HIP_G16_VERIFY_VM="0")
    out = subprocess.run([sys.executable, "-c", _G16_PATHS_CHILD, root, str(f1), str(f2)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lane = json.loads(out.stdout.strip().splitlines()[-1])
    assert got == lane
    assert got[0][:70] == [True] * 70 and got[0][150:153] == [True] * 3 and got[0][153] is False and got[2][:10] == [True] * 10
    assert got[1][:len(got[0])] == got[0] and not any(got[1][-8:])
    for i in list(range(66, 78)) + list(range(150, 162)):
        assert got[1][i] == g.verify_equality_with_commitment(blobs[i], blobs[i][266:], SS), i


def test_groth16_machine_batch_sizes(hip):
    """Batch sizes around the machine's 32-envelope workgroups (1, 2, 31, 33, 65, 95): every verdict right, valid and tampered envelopes
    interleaved (lanes past the end of a batch must neither be read nor written)."""
    import os
    import libzkp_amd as z
    import libzkp_amd.api as api
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        api.install_proving_key(kind, open(os.path.join(gold, name), "rb").read())
    rng = np.random.default_rng(123)
    vals = [int(x) for x in rng.integers(0, 2**63, 95, dtype=np.uint64)]
    proofs = z.prove_equality_batch(vals, vals)
    for n in (1, 2, 31, 33, 65, 95):
        blobs, want = [], []
        for i in range(n):
            if i % 3 == 1:
                b = bytearray(proofs[i]); b[20 + (7 * i) % 250] ^= 0x10; blobs.append(bytes(b)); want.append(False)
            else:
                blobs.append(proofs[i]); want.append(True)
        assert api._verify_snark_envelopes(0, blobs) == want, n


_G16_BATCH_CHILD = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
import libzkp_amd.api as api
gold = os.path.join(sys.argv[1], "tests", "golden")
for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
    api.install_proving_key(kind, open(os.path.join(gold, name), "rb").read())
out = []
for kind, hexes in json.load(open(sys.argv[2])):
    try:
        out.append(api._verify_snark_envelopes(kind, [bytes.fromhex(h) for h in hexes]))
    except Exception as e:
        out.append("error: " + str(e))
print(json.dumps(out))
"""


def test_groth16_one_pairing_check_per_batch_gives_the_per_envelope_verdicts(hip, tmp_path):
    """The batch check (g16_rlc.h; by default for batches of 8192 envelopes and more, here forced for every size in child processes):
    with ZKP_HIP_G16_BATCH_VERIFY_ONLY (no second, per-envelope pass) all-valid batches of 1 / 33 / 150 equality and 12 membership
    envelopes are accepted by the single check and every batch with a tampered or special envelope is refused by it; without that
    switch the verdicts of every batch equal the per-envelope path's, bad envelopes named one by one."""
    import json, os, subprocess, sys
    import libzkp_amd as z
    import libzkp_amd.api as api
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    gold = os.path.join(root, "tests", "golden")
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        api.install_proving_key(kind, open(os.path.join(gold, name), "rb").read())
    rng = np.random.default_rng(404)
    vals = [int(x) for x in rng.integers(0, 2**63, 150, dtype=np.uint64)]
    proofs = z.prove_equality_batch(vals, vals)
    sets = [[int(x) for x in rng.integers(0, 2**64, int(rng.integers(1, 65)), dtype=np.uint64)] for _ in range(12)]
    sets[3] = [0, 2**64 - 1, 5]
    mp = z.prove_membership_batch([s[i % len(s)] for i, s in enumerate(sets)], sets)
    flipped = list(proofs[:40])
    for i in (0, 7, 39):
        b = bytearray(flipped[i]); b[30 + i] ^= 2; flipped[i] = bytes(b)
    swapped = list(proofs[:33]); swapped[5] = proofs[5][:266] + proofs[6][266:]              # a valid proof under another commitment
    inf = list(proofs[:9]); b = bytearray(inf[4]); b[10:74] = bytes(63) + b"\x40"; inf[4] = bytes(b)
    short = list(proofs[:5]); short[2] = short[2][:-1]
    mbad = list(mp); b = bytearray(mbad[8]); b[14] ^= 1; mbad[8] = bytes(b)                    # a set element
    good = [(0, proofs[:1]), (0, proofs[:33]), (0, proofs), (1, mp), (1, mp[:1])]
    bad = [(0, flipped), (0, swapped), (0, inf), (0, short), (1, mbad)]
    f = tmp_path / "batches.json"
    f.write_text(json.dumps([[k, [e.hex() for e in b]] for k, b in good + bad]))

    def child(extra):
        env = dict(os.environ, ZKP_HIP_G16_BATCH_VERIFY_MIN="1", **extra)
        out = subprocess.run([sys.executable, "-c", _G16_BATCH_CHILD, root, str(f)], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])
    only = child({"ZKP_HIP_G16_BATCH_VERIFY_ONLY": "1"})
    for (k, b), got in zip(good, only[:len(good)]):
        assert got == [True] * len(b), (k, len(b), got if isinstance(got, str) else got.count(False))
    both = child({})
    want = [api._verify_snark_envelopes(k, b) for k, b in good + bad]                         # this process: batches this small take the per-envelope path
    assert both == want
    # tampering that leaves every point a valid group element reaches the pairing product and must fail the single check; a flipped bit
    # that breaks an encoding, or a refused header, removes that envelope alone (no pairing work) and the check of the others stands
    for got in only[len(good) + 1:len(good) + 3] + only[-1:]:
        assert isinstance(got, str) and "batch check did not stand" in got, got
    assert only[len(good) + 3] == [True, True, False, True, True] == want[len(good) + 3]
    assert only[len(good)] == want[len(good)] or (isinstance(only[len(good)], str) and "batch check did not stand" in only[len(good)])
    assert want[5].count(False) == 3 and want[6].count(False) == 1 and want[7].count(False) == 1 and want[9].count(False) == 1
    # default switches, this process: 8200 envelopes take the batch check (threshold 8193), and one envelope under its neighbour's commitment
    # sends the batch to the per-envelope pass, which names it
    big = [proofs[i % 150] for i in range(8200)]
    assert api._verify_snark_envelopes(0, big) == [True] * 8200
    big[5000] = big[5000][:266] + big[5001][266:]
    got = api._verify_snark_envelopes(0, big)
    assert got.count(False) == 1 and got[5000] is False
    # nothing but refused envelopes: no live envelope, the virtual envelope has no points -- the call still answers (all rejected)
    assert api._verify_snark_envelopes(0, [bytes(298)] * 8200) == [False] * 8200
    mixed = [proofs[i % 150] if i % 2 else bytes(298) for i in range(8200)]
    assert api._verify_snark_envelopes(0, mixed) == [bool(i % 2) for i in range(8200)]
