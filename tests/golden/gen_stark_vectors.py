"""Generates tests/golden/stark_oracle_vectors.json from the Python oracle (oracle/py/stark.py).

These are ORACLE outputs (parity with the reference's Winterfell is unpinned, see the oracle's header): they pin the
restatement against accidental change and give the GPU tests full-envelope known answers without running the oracle.
Run from the repo root:  python tests/golden/gen_stark_vectors.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.py import stark  # noqa: E402

CASES = [(0, 1), (30, 50), (100, 250), (5, 2**63), (0, 2**64 - 1), (2**64 - 2, 2**64 - 1), (123456789, 987654321), (7, 14), (1, 2**32)]


def main():
    out = {"note": "improvement-proof envelopes of oracle/py/stark.py: length, SHA-256, first 64 and last 32 bytes",
           "field_two_adic_root": str(stark.TWO_ADIC_ROOT), "vectors": []}
    for old, new in CASES:
        env = stark.prove_improvement(old, new)
        assert stark.verify_improvement(env, old)
        out["vectors"].append({"old": str(old), "new": str(new), "len": len(env), "sha256": hashlib.sha256(env).hexdigest(),
                               "head": env[:64].hex(), "tail": env[-32:].hex()})
    with open(os.path.join(ROOT, "tests", "golden", "stark_oracle_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
