"""Generates tests/golden/ristretto_libsodium.json with libsodium 1.0.18 (an independent ristretto255
implementation present in the build container at /opt/conda/lib/libsodium.so.23).  Fixture data only:
inputs and expected outputs.  Run from the repo root:  python tests/golden/gen_ristretto_libsodium.py
"""
import ctypes
import hashlib
import json
import os

so = ctypes.CDLL("/opt/conda/lib/libsodium.so.23")
so.sodium_init()
L = 2**252 + 27742317777372353535851937790883648493


def H(tag, i):
    return hashlib.sha512(b"libzkp-amd fixture %s %d" % (tag, i)).digest()


cases = []
for i in range(24):
    h1, h2 = H(b"p", i), H(b"q", i)
    p, q, s, d, m = (ctypes.create_string_buffer(32) for _ in range(5))
    so.crypto_core_ristretto255_from_hash(p, h1)
    so.crypto_core_ristretto255_from_hash(q, h2)
    so.crypto_core_ristretto255_add(s, p, q)
    so.crypto_core_ristretto255_sub(d, p, q)
    k = (int.from_bytes(H(b"k", i)[:32], "little") % (L - 1)) + 1
    so.crypto_scalarmult_ristretto255(m, k.to_bytes(32, "little"), p.raw)
    cases.append({"hash_p": h1.hex(), "hash_q": h2.hex(), "p": p.raw.hex(), "q": q.raw.hex(), "p_plus_q": s.raw.hex(),
                  "p_minus_q": d.raw.hex(), "k": k.to_bytes(32, "little").hex(), "k_times_p": m.raw.hex()})
base = []
for k in (1, 2, 3, 7, 2**64 - 1, L - 1):
    o = ctypes.create_string_buffer(32)
    so.crypto_scalarmult_ristretto255_base(o, k.to_bytes(32, "little"))
    base.append({"k": k.to_bytes(32, "little").hex(), "k_times_base": o.raw.hex()})
bad = []
for i in range(64):
    cand = bytearray(hashlib.sha256(b"bad %d" % i).digest())
    cand[31] &= 0x7F   # libsodium 1.0.18's canonicity check ignores bit 255 (RFC 9496 and dalek reject it): keep it clear
    cand = bytes(cand)
    bad.append({"bytes": cand.hex(), "valid": bool(so.crypto_core_ristretto255_is_valid_point(cand))})
out = {"generator": "libsodium 1.0.18 (/opt/conda/lib/libsodium.so.23)",
       "from_hash_add_sub_mul": cases, "base_multiples": base, "validity": bad}
with open(os.path.join(os.path.dirname(__file__), "ristretto_libsodium.json"), "w") as f:
    json.dump(out, f, indent=1)
print("wrote", len(cases), "cases")
