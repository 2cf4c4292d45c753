"""Byte formats around the proving path and the batched verification front ends (row N2):

  Proof envelope parsing            /root/reference/src/proof/mod.rs:38-83  (limits: utils/limits.rs)
  CompositeProof ("COMP" container) /root/reference/src/utils/composition.rs:31-331
  create_composite_proof, verify_composite_proof[_integrity_only], create_proof_with_metadata, extract_proof_metadata
                                    /root/reference/src/advanced/composite.rs:10-59
  validate_proof_chain, get_proof_info            /root/reference/src/advanced/mod.rs:224-247
  verify_proofs_parallel, verify_proof_cryptographic
                                    /root/reference/src/utils/performance.rs:251-293, utils/proof_helpers.rs:156-247

Pure host-side framing (SHA-256, length-prefixed fields) around proofs that the GPU produces and verifies; the
cryptographic checks go through libzkp_amd.api's batched GPU verifiers (Bulletproofs family, STARK, Groth16 pairing check).
Error mapping as in api.py: InvalidInput -> ValueError, InvalidProofFormat -> TypeError (error_handling.rs:39-50)."""
import hashlib

MAX_PROOF_TOTAL_BYTES = 1 << 20
MAX_PROOF_PAYLOAD_BYTES = 900 * 1024
MAX_COMMITMENT_BYTES = 256
MAX_COMPOSITE_PROOF_BYTES = 4 << 20
PROOF_VERSION = 2
SCHEME_BY_NAME = {"range": 1, "equality": 2, "threshold": 3, "membership": 4, "improvement": 5, "consistency": 6}


class ProofFormatError(TypeError):
    """ZkpError::InvalidProofFormat"""


def parse_proof(data):
    """Proof::from_bytes: (version, scheme, proof, commitment)"""
    data = bytes(data)
    if len(data) > MAX_PROOF_TOTAL_BYTES:
        raise ProofFormatError("Invalid proof format: proof too large: max %d bytes" % MAX_PROOF_TOTAL_BYTES)
    if len(data) < 10:
        raise ProofFormatError("Invalid proof format: proof too short for header")
    plen, clen = int.from_bytes(data[2:6], "little"), int.from_bytes(data[6:10], "little")
    if plen > MAX_PROOF_PAYLOAD_BYTES or clen > MAX_COMMITMENT_BYTES:
        raise ProofFormatError("Invalid proof format: proof or commitment payload exceeds limit")
    if len(data) != 10 + plen + clen:
        raise ProofFormatError("Invalid proof format: proof byte length mismatch")
    return data[0], data[1], data[10:10 + plen], data[10 + plen:]


def _composition_hash(proofs, metadata):
    h = hashlib.sha256(b"COMPOSITE_PROOF:" + len(proofs).to_bytes(4, "little"))
    for p in proofs:
        h.update(p)
    for k in sorted(metadata):
        kb = k.encode("utf-8")
        h.update(len(kb).to_bytes(4, "little") + kb + len(metadata[k]).to_bytes(4, "little") + metadata[k])
    return h.digest()


def _serialize_composite(proofs, metadata):
    out = b"COMP" + len(proofs).to_bytes(4, "little") + len(metadata).to_bytes(4, "little")
    for p in proofs:
        out += len(p).to_bytes(4, "little") + p
    for k, v in metadata.items():                      # the reference iterates a HashMap here: any order is a valid encoding
        kb = k.encode("utf-8")
        out += len(kb).to_bytes(4, "little") + kb + len(v).to_bytes(4, "little") + v
    return out + _composition_hash(proofs, metadata)


def parse_composite(data):
    """CompositeProof::from_bytes: (list of proof envelopes, metadata dict); raises ProofFormatError"""
    data = bytes(data)
    if len(data) > MAX_COMPOSITE_PROOF_BYTES:
        raise ProofFormatError("Invalid proof format: composite proof too large: max %d bytes" % MAX_COMPOSITE_PROOF_BYTES)
    if len(data) < 12:
        raise ProofFormatError("Invalid proof format: composite proof too short: expected at least 12 bytes, got %d" % len(data))
    if data[:4] != b"COMP":
        raise ProofFormatError("Invalid proof format: invalid composite proof header: expected 'COMP', got '%r'" % list(data[:4]))
    nproofs, nmeta = int.from_bytes(data[4:8], "little"), int.from_bytes(data[8:12], "little")
    if nproofs > 1000 or nmeta > 1000:
        raise ProofFormatError("Invalid proof format: composite proof has too many items: proofs=%d, metadata=%d" % (nproofs, nmeta))
    off, proofs, metadata = 12, [], {}
    for _ in range(nproofs):
        if off + 4 > len(data):
            raise ProofFormatError("Invalid proof format: truncated proof length")
        n = int.from_bytes(data[off:off + 4], "little"); off += 4
        if off + n > len(data):
            raise ProofFormatError("Invalid proof format: truncated proof data")
        parse_proof(data[off:off + n])
        proofs.append(data[off:off + n]); off += n
    for i in range(nmeta):
        if off + 4 > len(data):
            raise ProofFormatError("Invalid proof format: truncated metadata header at index %d: offset=%d, data_len=%d" % (i, off, len(data)))
        kl = int.from_bytes(data[off:off + 4], "little"); off += 4
        if kl > 1024:
            raise ProofFormatError("Invalid proof format: metadata key too large at index %d: key_len=%d" % (i, kl))
        if off + kl > len(data):
            raise ProofFormatError("Invalid proof format: truncated metadata key at index %d: offset=%d, key_len=%d, data_len=%d" % (i, off, kl, len(data)))
        try:
            key = data[off:off + kl].decode("utf-8")
        except UnicodeDecodeError:
            raise ProofFormatError("Invalid proof format: invalid metadata key at index %d: non-utf8 bytes" % i) from None
        off += kl
        if off + 4 > len(data):
            raise ProofFormatError("Invalid proof format: truncated metadata value length at index %d: offset=%d, data_len=%d" % (i, off, len(data)))
        vl = int.from_bytes(data[off:off + 4], "little"); off += 4
        if vl > 65536:
            raise ProofFormatError("Invalid proof format: metadata value too large at index %d: value_len=%d" % (i, vl))
        if off + vl > len(data):
            raise ProofFormatError("Invalid proof format: truncated metadata value at index %d: offset=%d, value_len=%d, data_len=%d" % (i, off, vl, len(data)))
        metadata[key] = data[off:off + vl]; off += vl
    if off + 32 > len(data):
        raise ProofFormatError("Invalid proof format: missing composition hash")
    if off + 32 != len(data):
        raise ProofFormatError("Invalid proof format: trailing bytes after composition hash: %d extra byte(s)" % (len(data) - off - 32))
    if data[off:off + 32] != _composition_hash(proofs, metadata):
        raise ProofFormatError("Invalid proof format: composition hash mismatch")
    return proofs, metadata


def create_composite_proof(proof_list):
    if not proof_list:
        raise ValueError("proof list cannot be empty")
    proofs = [bytes(p) for p in proof_list]
    for p in proofs:
        parse_proof(p)
    return _serialize_composite(proofs, {})


def create_proof_with_metadata(proof_data, metadata):
    proof = bytes(proof_data)
    parse_proof(proof)
    return _serialize_composite([proof], {str(k): bytes(v) for k, v in metadata.items()})


def extract_proof_metadata(composite_bytes):
    return parse_composite(composite_bytes)[1]


def verify_composite_proof_integrity_only(composite_bytes):
    parse_composite(composite_bytes)                   # from_bytes already rejects a wrong digest; verify_integrity is then true
    return True


def verify_proof_cryptographic_batch(envelopes):
    """verify_proof_cryptographic for a list of envelopes, one batched GPU call per scheme (proof_helpers.rs:156-247)."""
    from . import api
    out = [False] * len(envelopes)
    groups = {1: [], 2: [], 3: [], 4: [], 5: [], 6: []}
    for i, env in enumerate(envelopes):
        try:
            version, scheme, payload, commitment = parse_proof(env)
        except ProofFormatError:
            continue
        if version != PROOF_VERSION:
            continue
        if scheme == 2 and len(commitment) == 32:
            groups[2].append((i, env))
        elif scheme == 4 and len(commitment) == 32 and len(payload) > 4:
            groups[4].append((i, env))
        elif scheme == 1 and len(payload) >= 20 and len(commitment) == 32:
            mn, mx = int.from_bytes(payload[:8], "little"), int.from_bytes(payload[8:16], "little")
            if mn <= mx:
                groups[1].append((i, env, mn, mx))
        elif scheme == 3 and len(payload) >= 12 and len(commitment) == 32:
            groups[3].append((i, env, int.from_bytes(payload[:8], "little")))
        elif scheme == 5 and len(payload) >= 16 and len(commitment) == 32:
            groups[5].append((i, env, int.from_bytes(payload[:8], "little")))
        elif scheme == 6:
            groups[6].append((i, env))
    if groups[1]:
        for (i, *_), ok in zip(groups[1], api.verify_range_batch([g[1] for g in groups[1]], [g[2] for g in groups[1]], [g[3] for g in groups[1]])):
            out[i] = ok
    if groups[2]:
        for (i, *_), ok in zip(groups[2], api._verify_snark_envelopes(0, [g[1] for g in groups[2]])):
            out[i] = ok
    if groups[4]:
        for (i, *_), ok in zip(groups[4], api._verify_snark_envelopes(1, [g[1] for g in groups[4]])):
            out[i] = ok
    if groups[3]:
        for (i, *_), ok in zip(groups[3], api.verify_threshold_batch([g[1] for g in groups[3]], [g[2] for g in groups[3]])):
            out[i] = ok
    if groups[5]:
        for (i, *_), ok in zip(groups[5], api.verify_improvement_batch([g[1] for g in groups[5]], [g[2] for g in groups[5]])):
            out[i] = ok
    if groups[6]:
        for (i, *_), ok in zip(groups[6], api.verify_consistency_batch([g[1] for g in groups[6]])):
            out[i] = ok
    return out


def verify_composite_proof(composite_bytes):
    proofs, _ = parse_composite(composite_bytes)
    return all(verify_proof_cryptographic_batch(proofs))


def verify_proofs_parallel(proofs):
    """[(proof bytes, type name)] -> [bool]: the type must match the envelope's scheme id (performance.rs:269-293)."""
    idx, envs, out = [], [], [False] * len(proofs)
    for i, (data, name) in enumerate(proofs):
        try:
            version, scheme, _, _ = parse_proof(data)
        except ProofFormatError:
            continue
        if version == PROOF_VERSION and SCHEME_BY_NAME.get(name) == scheme:
            idx.append(i); envs.append(bytes(data))
    for i, ok in zip(idx, verify_proof_cryptographic_batch(envs)):
        out[i] = ok
    return out


def validate_proof_chain(proof_chain):
    for b in proof_chain:
        try:
            parse_proof(b)
        except ProofFormatError:
            return False
    return True


def get_proof_info(proof_bytes):
    version, scheme, payload, commitment = parse_proof(proof_bytes)
    return {"version": version, "scheme": scheme, "proof_size": len(payload), "commitment_size": len(commitment)}
