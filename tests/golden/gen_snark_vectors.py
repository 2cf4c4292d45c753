"""Generates tests/golden/snark_oracle_vectors.json from the Python oracle (oracle/py/groth16.py): equality and
membership envelopes under the committed test keys (setup seed 0..31) and fixed per-proof seeds, as length + SHA-256 +
the 256 proof bytes.  ORACLE outputs (parity with the reference is unpinned: OsRng setup and OsRng r, s): they pin the
C restatement (oracle/c/groth16.c) and the GPU prover to the bigint model without running it (a membership proof
takes the Python model about a minute).  Run from the repo root:  python tests/golden/gen_snark_vectors.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.py import groth16 as g  # noqa: E402

SS = bytes(range(32))
EQ = [(0, 1), (42, 2), (2**64 - 1, 3), (123456789, 4)]
MEM = [(25, (10, 20, 25, 30), 5), (7, (7,), 6), (2**40, tuple(range(2**40 - 63, 2**40 + 1)), 7)]


def seed(k):
    return hashlib.sha256(b"snark vector" + bytes([k])).digest()


def main():
    out = {"note": "envelopes of oracle/py/groth16.py under tests/golden/*_pk.bin", "equality": [], "membership": []}
    for v, k in EQ:
        env = g.prove_equality(v, v, SS, seed(k))
        assert g.verify_equality_with_commitment(env, g.commit_value_snark(v), SS)
        out["equality"].append({"value": str(v), "seed": seed(k).hex(), "len": len(env), "sha256": hashlib.sha256(env).hexdigest(), "envelope": env.hex()})
    for v, s, k in MEM:
        env = g.prove_membership(v, list(s), SS, seed(k))
        assert g.verify_membership(env, list(s), SS)
        out["membership"].append({"value": str(v), "set": [str(x) for x in s], "seed": seed(k).hex(), "len": len(env),
                                  "sha256": hashlib.sha256(env).hexdigest(), "envelope": env.hex()})
    with open(os.path.join(ROOT, "tests", "golden", "snark_oracle_vectors.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
