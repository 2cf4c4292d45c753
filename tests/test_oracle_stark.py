"""The improvement-proof (STARK) oracle: pinned primitives, prover/verifier round trips, the reference's reject cases,
and the committed oracle vectors (no GPU)."""
import hashlib
import json
import os
import random

import pytest

from oracle.py import stark as s
from oracle.py.blake3 import blake3

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_blake3_known_answers():
    # BLAKE3 specification test vectors (input byte i = i mod 251) and the two classic strings
    assert blake3(b"").hex() == "af1349b9f5f9a1a6a0404dea36dcc9499bcb25c9adc112b7cc9a93cae41f3262"
    assert blake3(b"abc").hex() == "6437b3ac38465133ffb63b75273a8db548c558465d79db03fd359c6cd5bd9d85"
    assert blake3(bytes([0])).hex() == "2d3adedff11b61f14c886e35afa036736dcd87a74d27b5c1510225d0f592e213"
    assert blake3(bytes(i % 251 for i in range(1024))).hex() == "42214739f095a406f3fc83deb889744ac00df831c10daa55189b5d121c855af7"


def test_field_constants():
    assert s.P == 340282366920938463463374557953744961537 and s.P == 2**128 - 45 * 2**40 + 1
    # multiplicative generator 3: order p-1 = 2^40 * 3 * 5 * ... ; 3^((p-1)/q) != 1 for the small prime factors
    for q in (2, 3, 5):
        assert pow(3, (s.P - 1) // q, s.P) != 1
    assert s.TWO_ADIC_ROOT == 23953097886125630542083529559205016746          # winter-math f128 TWO_ADIC_ROOT_OF_UNITY
    assert pow(s.TWO_ADIC_ROOT, 1 << 39, s.P) == s.P - 1
    g = s.root_of_unity(8)
    assert pow(g, 8, s.P) == 1 and pow(g, 4, s.P) != 1


def test_vint_round_trip():
    for v in [0, 1, 127, 128, 16383, 16384, 2**21 - 1, 2**21, 2**56 - 1, 2**56, 2**64 - 1]:
        b = s.vint(v)
        assert s.read_vint(b + b"x", 0) == (v, len(b))
    assert [len(s.vint(v)) for v in (0, 127, 128, 16383, 16384, 2**56 - 1, 2**56)] == [1, 1, 2, 2, 3, 8, 9]


def test_merkle_batch_openings():
    rnd = random.Random(4)
    leaves = [blake3(bytes([i])) for i in range(64)]
    tree = s.MerkleTree(leaves)
    for k in (1, 2, 5, 17, 32, 64):
        for _ in range(10):
            idx = sorted(rnd.sample(range(64), k))
            nodes = tree.prove_batch(idx)
            assert sum(len(x) for x in nodes) <= 32
            assert s.batch_root([leaves[i] for i in idx], idx, nodes, 6) == tree.root
            if k < 64:
                wrong = [leaves[i] for i in idx]
                wrong[0] = blake3(b"x")
                assert s.batch_root(wrong, idx, nodes, 6) != tree.root


def test_round_trip_and_reference_reject_cases():
    rnd = random.Random(8)
    cases = [(0, 1), (30, 50), (0, 2**64 - 1), (2**64 - 2, 2**64 - 1)] + [tuple(sorted((rnd.randrange(2**63), 2**63 + rnd.randrange(2**63)))) for _ in range(6)]
    for old, new in cases:
        d = {}
        proof = s.prove(old, new, d)
        assert len(proof) <= 3469 and len(d["positions"]) <= 32
        assert s.verify(proof, old, new)
        assert not s.verify(proof, old, new + 1 if new < 2**64 - 1 else new - 1)      # stark.rs:262-265: wrong public input
        assert not s.verify(proof, old + 1, new) or old + 1 == new
        env = s.prove_improvement(old, new)
        assert len(env) == 10 + 16 + len(proof) + 32 and env[:2] == bytes([2, 5])
        assert s.verify_improvement(env, old) and not s.verify_improvement(env, old + 1)
        bad = bytearray(env); bad[12] ^= 1                                              # tests/integration.rs:78-85 byte-12 tamper
        assert not s.verify_improvement(bytes(bad), old)
    with pytest.raises(ValueError, match="new value must be greater than old value"):
        s.prove_improvement(5, 5)


def test_every_byte_is_bound():
    rnd = random.Random(9)
    proof = s.prove(100, 250)
    for i in range(rnd.randrange(4), len(proof), 4):
        bad = bytearray(proof)
        bad[i] ^= 1 << rnd.randrange(8)
        assert not s.verify(bytes(bad), 100, 250), i


def test_non_linear_trace_is_rejected():
    """soundness of the restated verifier: a prover that commits to a trace violating next = cur + step cannot pass"""
    orig = s.trace_column
    try:
        def crooked(old, new):
            col, step = orig(old, new)
            col[3] = (col[3] + 1) % s.P
            return col, step
        s.trace_column = crooked
        with pytest.raises(AssertionError):          # the composition no longer fits one column ...
            s.prove(10, 500)
    finally:
        s.trace_column = orig


def test_committed_oracle_vectors():
    gold = json.load(open(os.path.join(GOLD, "stark_oracle_vectors.json")))
    assert int(gold["field_two_adic_root"]) == s.TWO_ADIC_ROOT
    for v in gold["vectors"]:
        env = s.prove_improvement(int(v["old"]), int(v["new"]))
        assert len(env) == v["len"] and hashlib.sha256(env).hexdigest() == v["sha256"]
        assert env[:64].hex() == v["head"] and env[-32:].hex() == v["tail"]
