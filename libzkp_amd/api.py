"""Host-side mirror of the reference's operator interface for the hot path.

Same names, argument meaning and error behaviour as the reference:
  prove_range                 /root/reference/src/proof/range_proof.rs:10-27   (validation.rs:5-18 messages)
  create_proof_batch .. process_batch, get_batch_status, clear_batch
                              /root/reference/src/advanced/batch.rs:35-175,262-283
  benchmark_proof_generation  /root/reference/src/advanced/mod.rs:83-172,204-215
Error mapping follows /root/reference/src/utils/error_handling.rs:39-50: InvalidInput -> ValueError(msg);
everything else -> RuntimeError("<Display prefix>: msg").
"""
import ctypes
import secrets
import threading
import time

import numpy as np

from . import _native, perf, batch_store

U64_MAX = 2**64 - 1


class ZkpBackendError(RuntimeError):
    pass


def _check_u64(name, x):
    if not isinstance(x, (int, np.integer)) or isinstance(x, bool) or x < 0 or x > U64_MAX:
        raise OverflowError("%s out of range for u64" % name)
    return int(x)


def _u64_array(name, xs):
    """u64 column of a batch.  Unsigned numpy arrays are taken as they are; anything else is checked element by element."""
    if isinstance(xs, np.ndarray) and xs.dtype.kind == "u" and xs.ndim == 1:
        return np.ascontiguousarray(xs, dtype=np.uint64)
    return np.array([_check_u64(name, x) for x in xs], dtype=np.uint64)


def _rows(blobs, cap=4096):
    """Byte strings of a batch as rows of one buffer (stride = the longest, at most `cap`: longer ones are rejected by the
    library through their recorded length).  Equal lengths -- the usual case -- are packed without a Python loop."""
    n = len(blobs)
    lens = np.fromiter((len(b) for b in blobs), dtype=np.uint32, count=n)
    stride = int(min(cap, max(16, int(lens.max()))))
    if int(lens.min()) == int(lens.max()) == stride:
        return np.frombuffer(b"".join(blobs), dtype=np.uint8).reshape(n, stride).copy(), lens, stride
    buf = np.zeros((n, stride), dtype=np.uint8)
    for i, b in enumerate(blobs):
        buf[i, : min(len(b), stride)] = np.frombuffer(b[:stride], dtype=np.uint8)
    return buf, lens, stride


def validate_range_params(value, mn, mx):
    """validation.rs:5-18."""
    if mn > mx:
        raise ValueError("min cannot be greater than max")
    if value < mn or value > mx:
        raise ValueError("value %d is not in range [%d, %d]" % (value, mn, mx))


def _P(a):
    return a.ctypes.data_as(ctypes.c_void_p)


BIT_WIDTHS = (8, 16, 32, 64)          # RangeProof::prove_single's valid bit sizes


def max_u64_for_bit_width(n_bits):
    """bulletproofs.rs:94-100"""
    return 2**64 - 1 if n_bits >= 64 else (1 << n_bits) - 1


def _check_bits(n_bits, what):
    if n_bits not in BIT_WIDTHS:         # upstream: ProofError::InvalidBitsize -> "... proof generation failed"
        raise ZkpBackendError("Backend error: %s proof generation failed (n_bits must be 8, 16, 32 or 64)" % what)
    return int(n_bits)


def prove_range_batch(values, mins, maxs, seeds=None, device=None, n_bits=64):
    """Batched prove_range / prove_range_with_bits: returns a list of proof bytes (one envelope per op; 1478 bytes for
    n_bits = 64, 128 fewer per halving of the width).

    Raises ValueError (reference message) if any op is invalid -- like process_batch, one failure fails the call.
    `seeds` (n x 32 bytes) pins the randomness tape; None draws fresh OS randomness like the reference.
    """
    n = len(values)
    n_bits = _check_bits(n_bits, "min range")
    v, mn, mx = _u64_array("value", values), _u64_array("min", mins), _u64_array("max", maxs)
    if not (len(mn) == n and len(mx) == n):
        raise ValueError("values, mins, maxs must have equal length")
    cap = np.uint64(max_u64_for_bit_width(n_bits))
    bad = (mn > mx) | (v < mn) | (v > mx)
    if bad.any():
        i = int(np.argmax(bad))
        validate_range_params(int(v[i]), int(mn[i]), int(mx[i]))                  # raises the reference's message
    if n and (((v - mn) > cap) | ((mx - v) > cap)).any():                          # bulletproofs.rs:121-129
        raise ZkpBackendError("Backend error: range width exceeds %d-bit capacity; use n_bits=64" % n_bits)
    if n == 0:
        return []
    L = _native.lib()
    if device is not None:
        _native.check(L.zkp_hip_init(int(device)), "zkp_hip_init")
    sp = None
    if seeds is not None:
        s = np.frombuffer(bytes(seeds), dtype=np.uint8) if not isinstance(seeds, np.ndarray) else seeds.astype(np.uint8).ravel()
        if s.size != 32 * n:
            raise ValueError("seeds must hold 32 bytes per op")
        s = np.ascontiguousarray(s)
        sp = _P(s)
    out = np.zeros((n, _native.RANGE_PROOF_BYTES), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint32)
    st = np.zeros(n, dtype=np.int32)
    rc = L.zkp_hip_prove_range_batch(n, _P(v), _P(mn), _P(mx), n_bits, sp, _P(out), _native.RANGE_PROOF_BYTES, _P(lens), _P(st))
    if rc < 0:
        raise ZkpBackendError("Backend error: %s" % _native.last_error())
    if rc > 0:
        bad = int(np.nonzero(st)[0][0])
        raise ZkpBackendError("Backend error: range proof generation failed for op %d (status %d)" % (bad, int(st[bad])))
    return [out[i, : lens[i]].tobytes() for i in range(n)]


def _seed_ptr(seeds, n):
    if seeds is None:
        return None, None
    s = np.frombuffer(bytes(seeds), dtype=np.uint8) if not isinstance(seeds, np.ndarray) else seeds.astype(np.uint8).ravel()
    if s.size != 32 * n:
        raise ValueError("seeds must hold 32 bytes per op")
    s = np.ascontiguousarray(s)
    return s, _P(s)


def _raise_backend(rc, st, what):
    if rc < 0:
        raise ZkpBackendError("Backend error: %s" % _native.last_error())
    if rc > 0:
        bad = int(np.nonzero(st)[0][0])
        raise ZkpBackendError("Backend error: %s failed for op %d (status %d)" % (what, bad, int(st[bad])))


def validate_threshold_params(values, threshold):
    """validation.rs:30-47."""
    if len(values) == 0:
        raise ValueError("values cannot be empty")
    total = sum(values)
    if total > U64_MAX:
        raise ValueError("integer overflow in sum calculation")
    if total < threshold:
        raise ValueError("sum %d is less than threshold %d" % (total, threshold))
    return total


def validate_consistency_params(data):
    """validation.rs:75-88."""
    if len(data) == 0:
        raise ValueError("data cannot be empty")
    if any(a > b for a, b in zip(data, data[1:])):
        raise ValueError("data is not monotonic non-decreasing")


def prove_threshold_batch(value_lists, thresholds, seeds=None, n_bits=64):
    """Batched prove_threshold / prove_threshold_with_bits (threshold_proof.rs:12-32): one envelope per op (762 bytes for
    n_bits = 64, 64 fewer per halving of the width)."""
    n = len(value_lists)
    n_bits = _check_bits(n_bits, "threshold range")
    lists = [[_check_u64("value", x) for x in vl] for vl in value_lists]
    thr = [_check_u64("threshold", t) for t in thresholds]
    for vl, t in zip(lists, thr):
        validate_threshold_params(vl, t)
        if sum(vl) - t > max_u64_for_bit_width(n_bits):                           # bulletproofs.rs:330-336
            raise ZkpBackendError("Backend error: sum - threshold exceeds %d-bit capacity; use n_bits=64" % n_bits)
    if n == 0:
        return []
    flat = np.array([x for vl in lists for x in vl], dtype=np.uint64)
    counts = np.array([len(vl) for vl in lists], dtype=np.uint32)
    th = np.array(thr, dtype=np.uint64)
    keep, sp = _seed_ptr(seeds, n)
    stride = 762
    out = np.zeros((n, stride), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint32)
    st = np.zeros(n, dtype=np.int32)
    rc = _native.lib().zkp_hip_prove_threshold_batch(n, _P(flat), _P(counts), _P(th), n_bits, sp, _P(out), stride, _P(lens), _P(st))
    _raise_backend(rc, st, "threshold proof generation")
    return [out[i, : lens[i]].tobytes() for i in range(n)]


def prove_consistency_batch(data_lists, seeds=None):
    """Batched prove_consistency (consistency_proof.rs:12-22)."""
    n = len(data_lists)
    lists = [[_check_u64("value", x) for x in dl] for dl in data_lists]
    for dl in lists:
        validate_consistency_params(dl)
    if n == 0:
        return []
    L = _native.lib()
    flat = np.array([x for dl in lists for x in dl], dtype=np.uint64)
    counts = np.array([len(dl) for dl in lists], dtype=np.uint32)
    stride = max(int(L.zkp_hip_consistency_proof_bytes(int(c))) for c in counts)
    keep, sp = _seed_ptr(seeds, n)
    out = np.zeros((n, stride), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint32)
    st = np.zeros(n, dtype=np.int32)
    rc = L.zkp_hip_prove_consistency_batch(n, _P(flat), _P(counts), sp, _P(out), stride, _P(lens), _P(st))
    _raise_backend(rc, st, "consistency proof generation")
    return [out[i, : lens[i]].tobytes() for i in range(n)]


def prove_threshold(values, threshold):
    return prove_threshold_batch([list(values)], [threshold])[0]


def prove_threshold_with_bits(values, threshold, n_bits):
    """threshold_proof.rs:17-32: sum(values) - threshold must fit in n_bits (8, 16, 32 or 64)."""
    return prove_threshold_batch([list(values)], [threshold], n_bits=n_bits)[0]


def prove_consistency(data):
    return prove_consistency_batch([list(data)])[0]


# ---------------------------------------------------------------- Groth16 (equality / membership), snark.rs
MAX_SET_SIZE = 64
_key_dir_override = None
_keys_loaded = {}
_reinstall = {}            # kind -> key bytes to load again on first use after shutdown()
_key_blobs = {}            # kind -> proving-key bytes of the loaded key (multi-GPU: rank 0's key is broadcast, sharding.py)
_KEY_PREFIX = {0: "equality_mimc", 1: "membership_mimc"}          # snark.rs:306,327
_snark_lock = threading.Lock()


def set_snark_key_dir(path):
    """snark.rs:141-170 (ConfigError -> TypeError, error_handling.rs:43-45).  Must precede the first SNARK proof."""
    global _key_dir_override
    if not path:
        raise TypeError("SNARK key directory cannot be empty")
    with _snark_lock:
        if _keys_loaded:
            raise TypeError("SNARK setup is already initialized; set LIBZKP_SNARK_KEY_DIR before first proof")
        if _key_dir_override is not None and _key_dir_override != path:
            raise TypeError("SNARK key directory already set to %s; new value %s rejected" % (_key_dir_override, path))
        _key_dir_override = path
    return True


def is_snark_setup_initialized():
    return bool(_keys_loaded)


def _ensure_key(kind):
    """load_or_generate_setup (snark.rs:122-139): loads `{dir}/{prefix}_pk.bin` (ark-serialize uncompressed
    ProvingKey<Bn254>, the reference's own key-file format) or runs a fresh setup on the GPU and persists it."""
    import os
    with _snark_lock:
        if kind in _keys_loaded:
            return
        d = _key_dir_override or os.environ.get("LIBZKP_SNARK_KEY_DIR")
        L = _native.lib()
        path = os.path.join(d, _KEY_PREFIX[kind] + "_pk.bin") if d else None
        if kind in _reinstall:                                      # after shutdown(): the process keeps ONE setup per circuit
            blob = _reinstall.pop(kind)
            if L.zkp_hip_groth16_load_key(kind, blob, len(blob)) != 0:
                raise ZkpBackendError("Configuration error: %s" % _native.last_error())
            _key_blobs[kind] = blob
            path = path or "<reinstalled>"
        elif path and os.path.exists(path):
            blob = open(path, "rb").read()
            if L.zkp_hip_groth16_load_key(kind, blob, len(blob)) != 0:
                raise ZkpBackendError("Configuration error: %s" % _native.last_error())
            _key_blobs[kind] = blob
        else:
            # load_or_generate_setup (snark.rs:122-139): fresh trusted setup (OS randomness), persisted if a key dir is set
            pk_len, vk_len = ctypes.c_uint64(), ctypes.c_uint64()
            cap_pk, cap_vk = 1 << 20, 1 << 16
            pk, vk = ctypes.create_string_buffer(cap_pk), ctypes.create_string_buffer(cap_vk)
            if L.zkp_hip_groth16_generate_key(kind, None, pk, cap_pk, ctypes.byref(pk_len), vk, cap_vk, ctypes.byref(vk_len)) != 0:
                raise ZkpBackendError("Proof generation failed: setup failed: %s" % _native.last_error())
            if d:
                try:
                    os.makedirs(d, exist_ok=True)
                    open(path, "wb").write(pk.raw[: pk_len.value])
                    open(os.path.join(d, _KEY_PREFIX[kind] + "_vk.bin"), "wb").write(vk.raw[: vk_len.value])
                except OSError:
                    pass                                              # the reference ignores persist errors too (snark.rs:131-133)
            _key_blobs[kind] = pk.raw[: pk_len.value]
            path = path or "<generated in memory>"
        _keys_loaded[kind] = path


def shutdown():
    """Release every device resource of the library (tables, keys, workspaces, streams).  The next call initialises again;
    the Groth16 keys of this process are kept on the host and reinstalled on first use, so proofs made before still verify."""
    with _snark_lock:
        _native.lib().zkp_hip_shutdown()
        _reinstall.update({k: _key_blobs[k] for k in _keys_loaded if k in _key_blobs})
        _keys_loaded.clear()


def export_proving_key(kind):
    """The loaded (or freshly generated) proving key of a circuit (0 equality, 1 membership) in ark-serialize form."""
    _ensure_key(kind)
    return _key_blobs[kind]


def install_proving_key(kind, blob):
    """Make `blob` THE proving key of this process (every rank of a multi-GPU job must prove under one setup)."""
    with _snark_lock:
        if _native.lib().zkp_hip_groth16_load_key(kind, blob, len(blob)) != 0:
            raise ZkpBackendError("Configuration error: %s" % _native.last_error())
        _key_blobs[kind] = bytes(blob)
        _keys_loaded[kind] = "<installed>"


def snark_commit_value_batch(values):
    v = np.array([_check_u64("value", x) for x in values], dtype=np.uint64)
    if len(v) == 0:
        return []
    out = np.zeros((len(v), 32), dtype=np.uint8)
    rc = _native.lib().zkp_hip_snark_commit_value_batch(len(v), _P(v), _P(out))
    if rc != 0:
        raise ZkpBackendError("Backend error: %s" % _native.last_error())
    return [out[i].tobytes() for i in range(len(v))]


def snark_commit_value(value):
    """python_api.rs:32 -> commitment.rs:14-16 (MiMC-5 over BN254 Fr, 32 bytes LE)."""
    return snark_commit_value_batch([value])[0]


def validate_membership_params(value, the_set):
    """validation.rs:50-63, 91-100."""
    if len(the_set) == 0:
        raise ValueError("set cannot be empty")
    if value not in the_set:
        raise ValueError("value %d is not in the provided set" % value)
    if len(the_set) > MAX_SET_SIZE:
        raise ValueError("set size %d exceeds maximum allowed size %d" % (len(the_set), MAX_SET_SIZE))


def prove_equality_batch(vals1, vals2, seeds=None):
    """Batched prove_equality (equality_proof.rs:10-32): one 298-byte envelope per op."""
    n = len(vals1)
    a = np.array([_check_u64("val1", x) for x in vals1], dtype=np.uint64)
    b = np.array([_check_u64("val2", x) for x in vals2], dtype=np.uint64)
    if len(b) != n:
        raise ValueError("vals1, vals2 must have equal length")
    if (a != b).any():
        raise ValueError("values are not equal")                     # validation.rs:21-27
    if n == 0:
        return []
    _ensure_key(0)
    keep, sp = _seed_ptr(seeds, n)
    out = np.zeros((n, 298), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint32)
    st = np.zeros(n, dtype=np.int32)
    rc = _native.lib().zkp_hip_prove_equality_batch(n, _P(a), _P(b), sp, _P(out), 298, _P(lens), _P(st))
    if rc != 0:
        raise ZkpBackendError("Proof generation failed: SNARK proof generation failed%s" % (": " + _native.last_error() if rc < 0 else ""))
    return [out[i, : lens[i]].tobytes() for i in range(n)]


def prove_membership_batch(values, sets, seeds=None):
    """Batched prove_membership (set_membership.rs:12-38)."""
    n = len(values)
    vs = [_check_u64("value", x) for x in values]
    ss = [[_check_u64("set element", x) for x in s] for s in sets]
    for v, s in zip(vs, ss):
        validate_membership_params(v, s)
    if n == 0:
        return []
    _ensure_key(1)
    flat = np.array([x for s in ss for x in s], dtype=np.uint64)
    counts = np.array([len(s) for s in ss], dtype=np.uint32)
    va = np.array(vs, dtype=np.uint64)
    stride = 10 + 4 + 8 * int(counts.max()) + 256 + 32
    keep, sp = _seed_ptr(seeds, n)
    out = np.zeros((n, stride), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint32)
    st = np.zeros(n, dtype=np.int32)
    rc = _native.lib().zkp_hip_prove_membership_batch(n, _P(va), _P(flat), _P(counts), sp, _P(out), stride, _P(lens), _P(st))
    if rc != 0:
        raise ZkpBackendError("Proof generation failed: SNARK membership proof generation failed%s" % (": " + _native.last_error() if rc < 0 else ""))
    return [out[i, : lens[i]].tobytes() for i in range(n)]


def prove_improvement_batch(olds, news):
    """Batched prove_improvement (improvement_proof.rs:10-35): Winterfell-style STARK envelopes (scheme 5); deterministic."""
    n = len(olds)
    o = np.array([_check_u64("old", x) for x in olds], dtype=np.uint64)
    w = np.array([_check_u64("new", x) for x in news], dtype=np.uint64)
    if len(w) != n:
        raise ValueError("olds, news must have equal length")
    if (w <= o).any():
        raise ValueError("new value must be greater than old value")        # validation.rs:63-71
    if n == 0:
        return []
    L = _native.lib()
    stride = int(L.zkp_hip_improvement_max_bytes())
    out = np.zeros((n, stride), dtype=np.uint8)
    lens = np.zeros(n, dtype=np.uint32)
    st = np.zeros(n, dtype=np.int32)
    rc = L.zkp_hip_prove_improvement_batch(n, _P(o), _P(w), _P(out), stride, _P(lens), _P(st))
    if rc != 0:
        raise ZkpBackendError("Proof generation failed: STARK proof generation failed%s" % (": " + _native.last_error() if rc < 0 else ""))
    return [out[i, : lens[i]].tobytes() for i in range(n)]


def prove_improvement(old, new):
    return prove_improvement_batch([old], [new])[0]


def prove_equality(val1, val2):
    return prove_equality_batch([val1], [val2])[0]


def prove_equality_advanced(val1, val2):
    """advanced/mod.rs:193-196: same semantics as prove_equality."""
    return prove_equality(val1, val2)


def prove_membership(value, set):  # noqa: A002
    return prove_membership_batch([value], [list(set)])[0]


def verify_range_batch(proofs, mins, maxs):
    """Batched verify_range (range_proof.rs:28-47): list of bools.  Never raises on malformed proofs (they are False)."""
    n = len(proofs)
    if len(mins) != n or len(maxs) != n:
        raise ValueError("proofs, mins, maxs must have equal length")
    if n == 0:
        return []
    mn, mx = _u64_array("min", mins), _u64_array("max", maxs)
    buf, lens, stride = _rows([bytes(p) for p in proofs])      # a range envelope is 1478 bytes: anything beyond 4096 is rejected outright
    ok = np.zeros(n, dtype=np.uint8)
    rc = _native.lib().zkp_hip_verify_range_batch(n, _P(buf), stride, _P(lens), _P(mn), _P(mx), _P(ok))
    _native.check(rc, "zkp_hip_verify_range_batch")
    return [x == 1 for x in ok]


def verify_threshold_batch(proofs, thresholds):
    """Batched verify_threshold (threshold_proof.rs:34-47): list of bools."""
    n = len(proofs)
    if len(thresholds) != n:
        raise ValueError("proofs, thresholds must have equal length")
    if n == 0:
        return []
    th = _u64_array("threshold", thresholds)
    buf, lens, stride = _rows([bytes(p) for p in proofs])
    ok = np.zeros(n, dtype=np.uint8)
    _native.check(_native.lib().zkp_hip_verify_threshold_batch(n, _P(buf), stride, _P(lens), _P(th), _P(ok)), "zkp_hip_verify_threshold_batch")
    return [x == 1 for x in ok]


def _verify_snark_envelopes(kind, blobs):
    """Groth16 pairing check of equality (kind 0) / membership (kind 1) envelopes under the loaded key; public inputs are the
    envelope's own commitment and embedded set."""
    n = len(blobs)
    if n == 0:
        return []
    _ensure_key(kind)
    buf, lens, stride = _rows(blobs, 4096)
    ok = np.zeros(n, dtype=np.uint8)
    fn = _native.lib().zkp_hip_verify_equality_batch if kind == 0 else _native.lib().zkp_hip_verify_membership_batch
    _native.check(fn(n, _P(buf), stride, _P(lens), _P(ok)), "zkp_hip_verify_%s_batch" % ("equality" if kind == 0 else "membership"))
    return ok.astype(bool).tolist()


def verify_equality_with_commitment_batch(proofs, commitments):
    """Batched verify_equality_with_commitment (equality_proof.rs:34-60): the envelope must carry exactly that commitment."""
    blobs = [bytes(p) for p in proofs]
    cms = [bytes(c) for c in commitments]
    if len(cms) != len(blobs):
        raise ValueError("proofs, commitments must have equal length")
    crypto = _verify_snark_envelopes(0, blobs)
    return [ok and len(c) == 32 and len(b) == 298 and b[266:] == c for ok, b, c in zip(crypto, blobs, cms)]


def verify_equality_with_commitment(proof, commitment):
    return verify_equality_with_commitment_batch([proof], [commitment])[0]


def verify_equality(proof, val1, val2):
    """equality_proof.rs:52-59"""
    val1, val2 = _check_u64("val1", val1), _check_u64("val2", val2)
    if val1 != val2:
        return False
    return verify_equality_with_commitment(proof, snark_commit_value(val1))


def verify_membership_batch(proofs, sets):
    """Batched verify_membership (set_membership.rs:40-70): the embedded set must equal `set` as a multiset."""
    blobs = [bytes(p) for p in proofs]
    if len(sets) != len(blobs):
        raise ValueError("proofs, sets must have equal length")
    crypto = _verify_snark_envelopes(1, blobs)
    out = []
    for ok, b, st in zip(crypto, blobs, sets):
        good = ok and len(b) >= 14
        if good:
            n = int.from_bytes(b[10:14], "little")
            emb = [int.from_bytes(b[14 + 8 * i: 22 + 8 * i], "little") for i in range(n)] if len(b) == 10 + 4 + 8 * n + 256 + 32 else None
            good = emb is not None and sorted(emb) == sorted(_check_u64("set element", x) for x in st)
        out.append(bool(good))
    return out


def verify_membership(proof, set):  # noqa: A002
    return verify_membership_batch([proof], [list(set)])[0]


def verify_consistency_batch(proofs):
    """Batched verify_consistency (consistency_proof.rs:24-32): list of bools."""
    n = len(proofs)
    if n == 0:
        return []
    blobs = [bytes(p) for p in proofs]
    buf, lens, stride = _rows(blobs, 1 << 20)
    ok = np.zeros(n, dtype=np.uint8)
    _native.check(_native.lib().zkp_hip_verify_consistency_batch(n, _P(buf), stride, _P(lens), _P(ok)), "zkp_hip_verify_consistency_batch")
    return ok.astype(bool).tolist()


def verify_consistency(proof):
    return verify_consistency_batch([proof])[0]


def verify_improvement_batch(proofs, olds):
    """Batched verify_improvement (improvement_proof.rs:37-68): list of bools."""
    n = len(proofs)
    if len(olds) != n:
        raise ValueError("proofs, olds must have equal length")
    if n == 0:
        return []
    ov = np.array([_check_u64("old", x) for x in olds], dtype=np.uint64)
    blobs = [bytes(p) for p in proofs]
    buf, lens, stride = _rows(blobs, 8192)
    ok = np.zeros(n, dtype=np.uint8)
    _native.check(_native.lib().zkp_hip_verify_improvement_batch(n, _P(buf), stride, _P(lens), _P(ov), _P(ok)), "zkp_hip_verify_improvement_batch")
    return ok.astype(bool).tolist()


def verify_improvement(proof, old):
    return verify_improvement_batch([proof], [old])[0]


def verify_threshold(proof, threshold):
    return verify_threshold_batch([proof], [threshold])[0]


def verify_range(proof, min, max):  # noqa: A002
    return verify_range_batch([proof], [min], [max])[0]


def prove_range(value, min, max):  # noqa: A002  (reference argument names)
    value, mn, mx = _check_u64("value", value), _check_u64("min", min), _check_u64("max", max)
    validate_range_params(value, mn, mx)
    return prove_range_batch([value], [mn], [mx])[0]


def prove_range_with_bits(value, min, max, n_bits):  # noqa: A002
    """range_proof.rs:14-27: value - min and max - value must both fit in n_bits (8, 16, 32 or 64)."""
    value, mn, mx = _check_u64("value", value), _check_u64("min", min), _check_u64("max", max)
    validate_range_params(value, mn, mx)
    return prove_range_batch([value], [mn], [mx], n_bits=n_bits)[0]


# ---------------------------------------------------------------- batch registry (batch.rs:18-175)
_registry = {}
_registry_lock = threading.Lock()


def _allocate_batch_id():
    while True:
        bid = secrets.randbits(64)
        if bid != 0 and bid not in _registry:
            return bid


def create_proof_batch():
    """batch.rs:36-48; with a store directory configured every change is written through (batch_store.rs:157-162)."""
    with _registry_lock:
        bid = _allocate_batch_id()
        _registry[bid] = []
        batch_store.persist_batch_if_configured(bid, _registry[bid])
        return bid


def _with_batch(batch_id, op):
    with _registry_lock:
        if batch_id not in _registry:
            raise ValueError("Invalid batch ID: %d" % batch_id)
        _registry[batch_id].append(op)
        batch_store.persist_batch_if_configured(batch_id, _registry[batch_id])


def open_batch_from_store(batch_id):
    """batch.rs:192-210: cold start from the on-disk store."""
    d = batch_store._store_dir_required()
    with _registry_lock:
        if batch_id in _registry:
            raise ValueError("batch %d is already open in this process" % batch_id)
        _registry[batch_id] = batch_store.read_batch_file(d, batch_id)


def refresh_batch_from_store(batch_id):
    """batch.rs:214-232: pick up what another process wrote."""
    d = batch_store._store_dir_required()
    with _registry_lock:
        if batch_id not in _registry:
            raise ValueError("batch %d is not loaded in this process" % batch_id)
        _registry[batch_id] = batch_store.read_batch_file(d, batch_id)


def export_batch_to_file(batch_id, dest):
    """batch.rs:236-245"""
    with _registry_lock:
        if batch_id not in _registry:
            raise ValueError("Invalid batch ID: %d" % batch_id)
        batch_store.export_proof_batch_to_path(_registry[batch_id], dest)


def import_batch_from_file(src):
    """batch.rs:249-260: new batch id; persisted if a store is configured."""
    ops = batch_store.import_proof_batch_from_path(src)
    with _registry_lock:
        bid = _allocate_batch_id()
        _registry[bid] = ops
        batch_store.persist_batch_if_configured(bid, ops)
        return bid


def batch_add_range_proof(batch_id, value, min, max):  # noqa: A002
    value, mn, mx = _check_u64("value", value), _check_u64("min", min), _check_u64("max", max)
    validate_range_params(value, mn, mx)
    _with_batch(batch_id, ("range", value, mn, mx))


def batch_add_equality_proof(batch_id, val1, val2):
    val1, val2 = _check_u64("val1", val1), _check_u64("val2", val2)
    if val1 != val2:
        raise ValueError("values are not equal")
    _with_batch(batch_id, ("equality", val1, val2))


def batch_add_threshold_proof(batch_id, values, threshold):
    values = [_check_u64("value", x) for x in values]
    validate_threshold_params(values, _check_u64("threshold", threshold))
    _with_batch(batch_id, ("threshold", tuple(values), threshold))


def batch_add_membership_proof(batch_id, value, set):  # noqa: A002
    value = _check_u64("value", value)
    the_set = [_check_u64("set element", x) for x in set]
    if not the_set:                                                  # batch.rs:94-97 validates membership only
        raise ValueError("set cannot be empty")
    if value not in the_set:
        raise ValueError("value %d is not in the provided set" % value)
    _with_batch(batch_id, ("membership", value, tuple(the_set)))


def batch_add_improvement_proof(batch_id, old, new):
    old, new = _check_u64("old", old), _check_u64("new", new)
    if new <= old:
        raise ValueError("new value must be greater than old value")
    _with_batch(batch_id, ("improvement", old, new))


def batch_add_consistency_proof(batch_id, data):
    data = [_check_u64("value", x) for x in data]
    validate_consistency_params(data)
    _with_batch(batch_id, ("consistency", tuple(data)))


def process_batch(batch_id, seeds=None):
    """batch.rs:110-140: consumes the batch (even if proving fails), returns proofs in insertion order."""
    with _registry_lock:
        if batch_id not in _registry:
            raise ValueError("Invalid batch ID: %d" % batch_id)
        ops = _registry.pop(batch_id)
    batch_store.delete_batch_file_if_configured(batch_id)            # batch.rs:120-121
    for op in ops:
        if op[0] == "membership" and len(op[2]) > MAX_SET_SIZE:      # set_membership.rs:14 (validate_set_size at prove time)
            raise ValueError("set size %d exceeds maximum allowed size %d" % (len(op[2]), MAX_SET_SIZE))
    return process_ops(ops, seeds)


def process_ops(ops, seeds=None):
    """The whole mixed batch through ONE C-ABI call (zkp_hip_process_batch, the compiled replacement of
    advanced::process_batch); same result as prove_ops, without the per-variant Python marshalling."""
    n = len(ops)
    if n == 0:
        return []
    seeds = None if seeds is None else bytes(seeds)
    if seeds is not None and len(seeds) != 32 * n:
        raise ValueError("seeds must hold 32 bytes per op")
    code = {"range": _native.OP_RANGE, "equality": _native.OP_EQUALITY, "threshold": _native.OP_THRESHOLD,
            "membership": _native.OP_MEMBERSHIP, "improvement": _native.OP_IMPROVEMENT, "consistency": _native.OP_CONSISTENCY}
    arr = (_native.Op * n)()
    lists = []
    cap = 0
    for i, o in enumerate(ops):
        k = o[0]
        e = arr[i]
        e.kind = code[k]
        if k == "range":
            e.a, e.b, e.c = o[1], o[2], o[3]; cap += 1478
        elif k == "equality":
            e.a, e.b = o[1], o[2]; cap += 298
            _ensure_key(0)
        elif k == "improvement":
            e.a, e.b = o[1], o[2]; cap += 3527
        elif k == "threshold":
            e.a, e.count, e.list_off = o[2], len(o[1]), len(lists); lists.extend(o[1]); cap += 762
        elif k == "membership":
            e.a, e.count, e.list_off = o[1], len(o[2]), len(lists); lists.extend(o[2]); cap += 10 + 4 + 8 * len(o[2]) + 256 + 32
            _ensure_key(1)
        else:
            e.count, e.list_off = len(o[1]), len(lists); lists.extend(o[1])
            cap += 10 + 4 + 32 * len(o[1]) + (4 + 672 + 32) * max(len(o[1]) - 1, 0) + 32
    la = np.array(lists if lists else [0], dtype=np.uint64)
    out = np.zeros(max(cap, 1), dtype=np.uint8)
    off = np.zeros(n + 1, dtype=np.uint64)
    st = np.zeros(n, dtype=np.int32)
    sp = None if seeds is None else ctypes.c_char_p(seeds)
    rc = _native.lib().zkp_hip_process_batch(n, ctypes.byref(arr), _P(la), sp, _P(out), cap, _P(off), _P(st))
    if rc < 0:
        raise ZkpBackendError("Backend error: %s" % _native.last_error())
    if rc != 0:
        i = int(np.nonzero(st)[0][0])
        raise ZkpBackendError("Proof generation failed: operation %d (%s) failed with status %d" % (i, ops[i][0], int(st[i])))
    raw = out.tobytes()
    return [raw[int(off[i]): int(off[i + 1])] for i in range(n)]


KINDS = ("range", "threshold", "consistency", "equality", "membership", "improvement")


def prove_kind(kind, sel, seeds=None):
    """One batched device call for ops of one variant (tuples as stored by batch_add_*)."""
    if kind == "range":
        return prove_range_batch([o[1] for o in sel], [o[2] for o in sel], [o[3] for o in sel], seeds=seeds)
    if kind == "threshold":
        return prove_threshold_batch([o[1] for o in sel], [o[2] for o in sel], seeds=seeds)
    if kind == "consistency":
        return prove_consistency_batch([o[1] for o in sel], seeds=seeds)
    if kind == "equality":
        return prove_equality_batch([o[1] for o in sel], [o[2] for o in sel], seeds=seeds)
    if kind == "membership":
        return prove_membership_batch([o[1] for o in sel], [list(o[2]) for o in sel], seeds=seeds)
    if kind == "improvement":
        return prove_improvement_batch([o[1] for o in sel], [o[2] for o in sel])
    raise ValueError("unsupported proof type: %s" % kind)


def prove_ops(ops, seeds=None, prover=prove_kind, select=None):
    """Bucket by variant (one batched device call each), then restore insertion order (batch.rs:123-131 is
    order-preserving).  `select(kind, idx)` may narrow each bucket to the indices this process should prove
    (multi-GPU sharding); unproved ops stay None."""
    seeds = None if seeds is None else bytes(seeds)
    if seeds is not None and len(seeds) != 32 * len(ops):
        raise ValueError("seeds must hold 32 bytes per op")
    out = [None] * len(ops)
    for kind in KINDS:
        idx = [i for i, o in enumerate(ops) if o[0] == kind]
        if select is not None:
            idx = select(kind, idx)
        if not idx:
            continue
        sd = None if seeds is None else b"".join(seeds[32 * i: 32 * i + 32] for i in idx)
        proofs = prover(kind, [ops[i] for i in idx], sd)
        for i, p in zip(idx, proofs):
            out[i] = p
    return out


def get_batch_status(batch_id):
    with _registry_lock:
        if batch_id not in _registry:
            raise ValueError("Invalid batch ID: %d" % batch_id)
        ops = list(_registry[batch_id])
    out = {"total_operations": len(ops)}
    for key, name in (("range_proofs", "range"), ("equality_proofs", "equality"), ("threshold_proofs", "threshold"),
                      ("membership_proofs", "membership"), ("improvement_proofs", "improvement"), ("consistency_proofs", "consistency")):
        out[key] = sum(1 for o in ops if o[0] == name)
    return out


def clear_batch(batch_id):
    with _registry_lock:
        _registry.pop(batch_id, None)
    batch_store.delete_batch_file_if_configured(batch_id)            # batch.rs:183-184


# ---------------------------------------------------------------- benchmark harness (advanced/mod.rs:83-172)
def benchmark_proof_generation_numeric(proof_type, iterations):
    runners = {"range": lambda: prove_range(50, 0, 100),                          # mod.rs:94-104 fixed inputs
               "threshold": lambda: prove_threshold([10, 20, 30, 40], 50),
               "consistency": lambda: prove_consistency([10, 20, 30, 40, 50]),
               "equality": lambda: prove_equality(42, 42),
               "membership": lambda: prove_membership(25, [10, 20, 25, 30, 40]),
               "improvement": lambda: prove_improvement(30, 50)}                     # mod.rs:100
    if proof_type not in runners:
        raise ValueError("unsupported proof type: %s" % proof_type)
    times = []
    for _ in range(iterations):
        t0 = time.perf_counter()
        runners[proof_type]()
        dt = time.perf_counter() - t0
        perf.METRICS.record_operation(proof_type + "_proof", dt)                  # mod.rs:113-123
        times.append(dt * 1e3)
    if not times:
        raise ValueError("no successful proof generations")
    total = sum(times)
    avg = total / len(times)
    var = sum((x - avg) ** 2 for x in times) / len(times)
    return {
        "iterations": float(iterations), "successful_iterations": float(len(times)), "success_rate": 100.0,
        "total_time_ms": total, "avg_time_ms": avg, "min_time_ms": min(times), "max_time_ms": max(times),
        "std_dev_ms": var ** 0.5, "proofs_per_second": len(times) / (total / 1e3), "throughput_ms_per_proof": total / len(times),
    }


# ---------------------------------------------------------------- cache / metrics entry points (advanced/mod.rs:25-80,175-191,218-221)
def clear_cache():
    perf.CACHE.clear()


def get_cache_stats():
    return {"size": perf.CACHE.size()}


def get_performance_metrics():
    return perf.performance_metrics()


def prove_range_cached(value, min, max):  # noqa: A002
    """advanced/mod.rs:175-191: a hit returns the cached proof bytes; a miss proves, records the timing and stores."""
    value, mn, mx = _check_u64("value", value), _check_u64("min", min), _check_u64("max", max)
    key = perf.generate_cache_key("range_proof", ("%d:%d:%d" % (value, mn, mx)).encode())
    hit = perf.CACHE.get(key)
    if hit is not None:
        return hit
    t0 = time.perf_counter()
    proof = prove_range(value, mn, mx)
    perf.METRICS.record_operation("range_proof", time.perf_counter() - t0)
    perf.CACHE.put(key, proof)
    return proof


def prove_threshold_optimized(values, threshold):
    """advanced/mod.rs:218-221: delegates to prove_threshold."""
    return prove_threshold(values, threshold)


def benchmark_proof_generation(proof_type, iterations):
    """mod.rs:204-215: the Python variant stringifies every value and adds proof_type."""
    res = {k: repr(float(v)) for k, v in benchmark_proof_generation_numeric(proof_type, iterations).items()}
    res["proof_type"] = proof_type
    return res
