"""Pins of the oracle's building blocks against independent sources (no GPU)."""
import hashlib
import os

from oracle.py import bulletproofs as bp
from oracle.py import merlin, ristretto as R


def test_keccak_against_hashlib():
    for n in (0, 1, 57, 58, 71, 72, 73, 135, 136, 137, 500):
        d = bytes((i * 131 + n) & 255 for i in range(n))
        assert merlin.shake256(d, 300) == hashlib.shake_256(d).digest(300)
        assert merlin.sha3_512(d) == hashlib.sha3_512(d).digest()


def test_merlin_published_kat():
    # merlin crate, transcript.rs `equivalence_simple`
    t = merlin.Transcript(b"test protocol")
    t.append_message(b"some label", b"some data")
    assert t.challenge_bytes(b"challenge", 32).hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


def test_ristretto_against_libsodium_fixture(golden_ristretto):
    for c in golden_ristretto["from_hash_add_sub_mul"]:
        p = R.from_uniform_bytes(bytes.fromhex(c["hash_p"]))
        q = R.from_uniform_bytes(bytes.fromhex(c["hash_q"]))
        assert p.encode().hex() == c["p"] and q.encode().hex() == c["q"]
        assert (p + q).encode().hex() == c["p_plus_q"] and (p - q).encode().hex() == c["p_minus_q"]
        k = int.from_bytes(bytes.fromhex(c["k"]), "little")
        assert (k * p).encode().hex() == c["k_times_p"]
        d = R.decode(bytes.fromhex(c["p"]))
        assert d is not None and d.encode().hex() == c["p"]
    for c in golden_ristretto["base_multiples"]:
        k = int.from_bytes(bytes.fromhex(c["k"]), "little")
        assert (k * R.BASEPOINT).encode().hex() == c["k_times_base"]
    for c in golden_ristretto["validity"]:
        assert (R.decode(bytes.fromhex(c["bytes"])) is not None) == c["valid"]


def test_rfc9496_basepoint_multiples():
    # RFC 9496 appendix A.1 (first multiples of the generator)
    exp = ["0000000000000000000000000000000000000000000000000000000000000000",
           "e2f2ae0a6abc4e71a884a961c500515f58e30b6aa582dd8db6a65945e08d2d76",
           "6a493210f7499cd17fecb510ae0cea23a110e8d5b901f8acadd3095c73a3b919"]
    for k, e in enumerate(exp):
        assert (R.IDENTITY if k == 0 else k * R.BASEPOINT).encode().hex() == e


def test_generators(golden_bp):
    # B_blinding value confirmed independently in SURVEY.md section 8c
    assert bp.B_BLINDING.encode().hex() == "8c9240b456a9e6dc65c377a1048d745f94a08cdb7f44cbcd7b46f34048871134"
    assert golden_bp["generators"]["0"] == R.BASEPOINT.encode().hex()
    G, H = bp.party_gens(64)
    assert golden_bp["generators"]["2"] == G[0].encode().hex() and golden_bp["generators"]["129"] == H[63].encode().hex()
    # the chain is prefix-stable: BulletproofGens::new(8, ..) party 0 is a prefix of new(64, ..)
    assert [p.encode() for p in bp.generators_chain(b"G\0\0\0\0", 8)] == [p.encode() for p in G[:8]]


def test_tape_is_one_shake_block():
    s = os.urandom(32)
    assert bp.draw64(s, 3, 9) == hashlib.shake_256(b"libzkp-amd/tape/v1" + s + (3).to_bytes(4, "little") + (9).to_bytes(4, "little")).digest(64)


def test_decode_rejects_noncanonical_and_negative():
    good = R.BASEPOINT.encode()
    assert R.decode(good) is not None
    hi = bytearray(good)
    hi[31] |= 0x80                      # bit 255 set: non-canonical (dalek / RFC 9496 reject)
    assert R.decode(bytes(hi)) is None
    assert R.decode((R.P + 2).to_bytes(32, "little")) is None     # s >= p
    assert R.decode((1).to_bytes(32, "little")) is None           # negative s
