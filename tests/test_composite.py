"""The "COMP" composite container, metadata wrapper and proof-info helpers (host-side byte formats, no GPU):
round trips, the reference's limits and error classes, digest binding (composition.rs, advanced/composite.rs)."""
import hashlib

import pytest

import libzkp_amd as z
from libzkp_amd import composite as c
from oracle.py import stark


def env(scheme, payload, commitment=bytes(32)):
    return bytes([2, scheme]) + len(payload).to_bytes(4, "little") + len(commitment).to_bytes(4, "little") + payload + commitment


def test_round_trip_and_digest_binding():
    p1, p2 = env(1, b"a" * 40), stark.prove_improvement(3, 9)
    comp = z.create_composite_proof([p1, p2])
    assert comp[:4] == b"COMP" and comp[4:12] == (2).to_bytes(4, "little") + (0).to_bytes(4, "little")
    assert comp[-32:] == hashlib.sha256(b"COMPOSITE_PROOF:" + (2).to_bytes(4, "little") + p1 + p2).digest()
    assert z.verify_composite_proof_integrity_only(comp) is True
    proofs, meta = c.parse_composite(comp)
    assert proofs == [p1, p2] and meta == {}
    bad = bytearray(comp); bad[20] ^= 1
    with pytest.raises(TypeError, match="composition hash mismatch|proof byte length mismatch|truncated"):
        z.verify_composite_proof_integrity_only(bytes(bad))
    with pytest.raises(TypeError, match="trailing bytes after composition hash: 1 extra"):
        z.verify_composite_proof_integrity_only(comp + b"\0")
    with pytest.raises(ValueError, match="proof list cannot be empty"):
        z.create_composite_proof([])
    with pytest.raises(TypeError, match="proof byte length mismatch"):
        z.create_composite_proof([p1[:-1]])


def test_metadata_wrapper():
    p = env(3, b"x" * 30)
    comp = z.create_proof_with_metadata(p, {"issuer": b"kyc-1", "b": b"", "a": b"\x01\x02"})
    assert z.extract_proof_metadata(comp) == {"issuer": b"kyc-1", "b": b"", "a": b"\x01\x02"}
    # the digest covers the metadata in sorted key order, whatever order the pairs were written in
    swapped = c._serialize_composite([p], {"a": b"\x01\x02", "issuer": b"kyc-1", "b": b""})
    assert swapped[-32:] == comp[-32:] and z.extract_proof_metadata(swapped) == z.extract_proof_metadata(comp)
    tam = comp.replace(b"kyc-1", b"kyc-2")
    with pytest.raises(TypeError, match="composition hash mismatch"):
        z.extract_proof_metadata(tam)
    with pytest.raises(TypeError, match="metadata value too large"):
        c.parse_composite(b"COMP" + (0).to_bytes(4, "little") + (1).to_bytes(4, "little") + (1).to_bytes(4, "little") + b"k" + (70000).to_bytes(4, "little") + bytes(70000 + 32))
    with pytest.raises(TypeError, match="too many items"):
        c.parse_composite(b"COMP" + (1001).to_bytes(4, "little") + (0).to_bytes(4, "little") + bytes(32))
    with pytest.raises(TypeError, match="too short"):
        c.parse_composite(b"COMP")
    with pytest.raises(TypeError, match="invalid composite proof header"):
        c.parse_composite(b"XOMP" + bytes(40))


def test_proof_info_and_chain():
    p = stark.prove_improvement(1, 2)
    info = z.get_proof_info(p)
    assert info == {"version": 2, "scheme": 5, "proof_size": len(p) - 42, "commitment_size": 32}
    assert z.validate_proof_chain([]) and z.validate_proof_chain([p, env(1, b"")]) and not z.validate_proof_chain([p, p[:-1]])
    with pytest.raises(TypeError, match="proof too short for header"):
        z.get_proof_info(b"\x02\x01")
    with pytest.raises(TypeError, match="payload exceeds limit"):
        z.get_proof_info(bytes([2, 1]) + (901 * 1024).to_bytes(4, "little") + (0).to_bytes(4, "little"))
