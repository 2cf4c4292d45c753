"""The Python model's prover against its verifier, on the reference's own accept/reject cases (no GPU).

Reference tests mirrored: bulletproofs.rs:700-705 (prove_range(5,0,10) verifies, fails for bounds (0,4)),
tests/integration.rs:78-85 (tamper byte 12 -> reject), wire round trip bulletproofs.rs:690-698.
"""
import pytest

from oracle.py import bulletproofs as bp
from oracle.py.merlin import Transcript

SEED = bytes(range(32))


def test_wire_roundtrip():
    body, commit = b"hello proof body", bytes([7]) * 32
    assert bp._unwire(bp._wire(body, commit)) == (body, commit)
    assert bp._unwire(bp._wire(body, commit) + b"x") is None


@pytest.mark.parametrize("n,v", [(8, 0), (8, 255), (16, 40000)])
def test_single_accept_reject(n, v):
    pr, V = bp.prove_single(Transcript(b"t"), v, 99, n, SEED, 0)
    assert len(pr) == 32 * (9 + 2 * (n.bit_length() - 1))
    assert bp.verify_single(Transcript(b"t"), pr, V, n)
    assert not bp.verify_single(Transcript(b"u"), pr, V, n)
    for pos in (0, 40, 130, 230, len(pr) - 1):
        bad = bytearray(pr)
        bad[pos] ^= 1
        assert not bp.verify_single(Transcript(b"t"), bytes(bad), V, n)


def test_out_of_range_value_cannot_be_proven():
    with pytest.raises(ValueError):
        bp.prove_single(Transcript(b"t"), 256, 1, 8, SEED, 0)


def test_golden_range_verifies_and_reference_negative_cases(golden_bp):
    for c in golden_bp["range"]:
        env = bytes.fromhex(c["proof"])
        assert len(env) == 1478 and env[0] == 2 and env[1] == 1
        wire = bp._wire(env[10:10 + 1436], env[10 + 1436:])
        assert bp.verify_range_with_bounds_bits(wire, c["min"], c["max"])
    c = golden_bp["range"][0]          # prove_range(50, 0, 100)
    env = bytes.fromhex(c["proof"])
    wire = bp._wire(env[10:10 + 1436], env[10 + 1436:])
    assert not bp.verify_range_with_bounds_bits(wire, 0, 99)
    bad = bytearray(env)
    bad[12] ^= 1                       # integration.rs:78-85
    assert not bp.verify_range_with_bounds_bits(bp._wire(bytes(bad[10:10 + 1436]), bytes(bad[10 + 1436:])), 0, 100)


def test_range_rejects_out_of_range():
    with pytest.raises(ValueError):
        bp.prove_range_with_bounds_bits(11, 0, 10, 64, SEED)
    with pytest.raises(ValueError):
        bp.prove_range_with_bounds_bits(300, 0, 1000, 8, SEED)
