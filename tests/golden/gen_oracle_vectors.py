"""Generates tests/golden/bulletproofs_oracle_vectors.json from the Python bigint model (oracle/py).
These are NOT reference outputs (the reference cannot be built here and its proofs are randomised);
they freeze this project's own tape-seeded answers so the C oracle, the emulated kernels and the HIP
path are all held to the same bytes from now on.  Run from the repo root.
"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.getcwd())
from oracle.py import bulletproofs as bp  # noqa: E402
from oracle.py.merlin import Transcript  # noqa: E402


def envelope(scheme, wire):
    body, commit = bp._unwire(wire)
    return bytes([2, scheme]) + len(body).to_bytes(4, "little") + len(commit).to_bytes(4, "little") + body + commit


def seed(i):
    return hashlib.sha256(b"libzkp-amd golden seed %d" % i).digest()


out = {"generators": {str(i): p.encode().hex() for i, p in enumerate([bp.B, bp.B_BLINDING] + bp.party_gens(64)[0] + bp.party_gens(64)[1])},
       "tape": [{"seed": seed(i).hex(), "proof_idx": i * 3, "slot": i * 11, "draw64": bp.draw64(seed(i), i * 3, i * 11).hex()} for i in range(4)],
       "single": [], "range": [], "threshold": [], "consistency": []}
for i, (label, v, bl, n) in enumerate([(b"libzkp_bulletproof", 200, 12345, 8), (b"libzkp_range_min", 2**64 - 1, bp.L - 1, 64)]):
    pr, V = bp.prove_single(Transcript(label), v, bl, n, seed(10 + i), 4)
    assert bp.verify_single(Transcript(label), pr, V, n)
    out["single"].append({"label": label.decode(), "v": v, "blinding": bl.to_bytes(32, "little").hex(), "n_bits": n, "seed": seed(10 + i).hex(),
                          "proof_idx": 4, "proof": pr.hex(), "commitment": V.hex()})
for i, (v, mn, mx) in enumerate([(50, 0, 100), (2**32, 0, 2**32), (7, 7, 7), (2**63 + 5, 3, 2**64 - 1)]):
    w = bp.prove_range_with_bounds_bits(v, mn, mx, 64, seed(20 + i))
    assert bp.verify_range_with_bounds_bits(w, mn, mx)
    out["range"].append({"value": v, "min": mn, "max": mx, "seed": seed(20 + i).hex(), "proof": envelope(1, w).hex()})
w = bp.prove_threshold_bits([10, 20, 30, 40], 50, 64, seed(30))
assert bp.verify_threshold(w, 50)
out["threshold"].append({"values": [10, 20, 30, 40], "threshold": 50, "seed": seed(30).hex(), "proof": envelope(3, w).hex()})
w = bp.prove_consistency([10, 20, 20, 50], seed(40))
assert bp.verify_consistency(w)
out["consistency"].append({"data": [10, 20, 20, 50], "seed": seed(40).hex(), "proof": envelope(6, w).hex()})
with open(os.path.join("tests", "golden", "bulletproofs_oracle_vectors.json"), "w") as f:
    json.dump(out, f, indent=1)
print("ok")
