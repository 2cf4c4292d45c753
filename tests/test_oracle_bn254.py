"""BN254 / Groth16 oracle pins (no GPU): pairing bilinearity, ark serialisation, MiMC, prove -> pairing-verify, the
reference's accept/reject cases (snark.rs:617-641), and the committed key fixture."""
import os

from oracle.py import bn254 as b
from oracle.py import groth16 as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SS = bytes(range(32))
SEED = bytes(range(1, 33))


def test_generators_and_pairing():
    assert b.G1C.mul_pt(b.G1, b.R, reduce=False) is None and b.G2C.mul_pt(b.G2, b.R, reduce=False) is None
    e1 = b.pairing(b.G2, b.G1)
    assert e1 != b.F12_ONE and b.f12_pow(e1, b.R) == b.F12_ONE
    a, c = 123456789, 987654321
    assert b.pairing(b.G2C.mul_pt(b.G2, c), b.G1C.mul_pt(b.G1, a)) == b.f12_pow(e1, a * c % b.R)
    assert b.pairing_product_is_one([(b.G1, b.G2), (b.G1C.neg_pt(b.G1), b.G2)])


def test_ark_point_serialisation_roundtrip_and_flags():
    for k in (1, 2, 77, b.R - 1):
        p, q = b.G1C.mul_pt(b.G1, k), b.G2C.mul_pt(b.G2, k)
        assert b.de_g1(b.ser_g1(p)) == (True, p) and b.de_g2(b.ser_g2(q)) == (True, q)
    assert b.ser_g1(None)[63] == 0x40 and b.de_g1(b.ser_g1(None)) == (True, None)
    p = b.G1C.mul_pt(b.G1, 5)
    neg = b.G1C.neg_pt(p)
    assert (b.ser_g1(p)[63] & 0x80) != (b.ser_g1(neg)[63] & 0x80)        # exactly one of {y, -y} is "larger"
    bad = bytearray(b.ser_g1(p))
    bad[0] ^= 1
    assert b.de_g1(bytes(bad))[0] is False                                # off-curve


def test_mimc_matches_committed_vectors_and_reference_properties():
    import json
    vec = json.load(open(os.path.join(ROOT, "tests", "golden", "groth16_vectors.json")))
    for k, h in vec["mimc"].items():
        assert g.commit_value_snark(int(k)).hex() == h
    assert g.mimc_hash_native(42) == g.mimc_hash_native(42) != g.mimc_hash_native(43)      # snark.rs:617-622
    assert all(c < b.R for c in g.mimc_constants()) and len(set(g.mimc_constants())) == 110
    assert pow(5, (b.R - 1) // 512, b.R) != 1 and pow(pow(5, (b.R - 1) // 512, b.R), 512, b.R) == 1


def test_circuit_shapes():
    cs = g.equality_circuit(42, 42, g.mimc_hash_native(42))
    assert (len(cs.rows), cs.n_inst, cs.n_wit, g.domain_size(cs), g.is_satisfied(cs)) == (332, 2, 332, 512, True)
    assert not g.is_satisfied(g.equality_circuit(42, 43, g.mimc_hash_native(42)))
    assert not g.is_satisfied(g.equality_circuit(42, 42, g.mimc_hash_native(41)))
    sel, sv, ir = g.membership_inputs(25, [10, 20, 25, 30, 40])
    cs = g.membership_circuit(25, sel, sv, ir, g.mimc_hash_native(25))
    assert (len(cs.rows), cs.n_inst, cs.n_wit, g.domain_size(cs), g.is_satisfied(cs)) == (653, 130, 523, 1024, True)
    sel2 = [False] * 64
    sel2[5] = True                                                        # selects a padding slot
    assert not g.is_satisfied(g.membership_circuit(25, sel2, sv, ir, g.mimc_hash_native(25)))


def test_groth16_equality_roundtrip_like_the_reference():
    """snark.rs:630-641: prove(42,42) verifies; a wrong commitment is rejected.  Also: the MSM/FFT prover and the
    toxic-waste prover give the same bytes, and the committed key fixture is this key."""
    key = g.equality_key(SS)
    env = g.prove_equality(42, 42, SS, SEED)
    assert len(env) == 298 and env[:2] == bytes([2, 2])
    assert g.verify_equality_with_commitment(env, g.commit_value_snark(42), SS)
    wrong = g.commit_value_snark(99)
    assert not g.verify_equality_with_commitment(env[:266] + wrong, wrong, SS)
    cs = g.equality_circuit(42, 42, g.mimc_hash_native(42))
    assert g.prove_with_trapdoor(key, cs, g.draw_fr(SEED, 0x47313600, 0), g.draw_fr(SEED, 0x47313600, 1)) == env[10:266]
    tampered = bytearray(env)
    tampered[12] ^= 1                                                     # tests/integration.rs:78-85
    assert not g.verify_equality_with_commitment(bytes(tampered), g.commit_value_snark(42), SS)
    assert g.serialize_pk(key) == open(os.path.join(ROOT, "tests", "golden", "equality_mimc_pk.bin"), "rb").read()
    h = g.witness_map(cs)
    assert h[-1] == 0
