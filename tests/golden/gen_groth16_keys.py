"""Generates the Groth16 proving keys used by the tests from the Python oracle (oracle/py/groth16.py) with the fixed
setup seed below, in ark-serialize uncompressed ProvingKey format -- the same byte layout as the reference's
`equality_mimc_pk.bin` / `membership_mimc_pk.bin` key files (snark.rs:31-38).  These are this project's own test keys
(toxic waste derived from a public seed: for tests only), not reference outputs.  Also writes MiMC known answers.
Run from the repo root: python tests/golden/gen_groth16_keys.py"""
import json
import os
import sys

sys.path.insert(0, os.getcwd())
from oracle.py import groth16 as g  # noqa: E402

SETUP_SEED = bytes(range(32))
out = os.path.join("tests", "golden")
open(os.path.join(out, "equality_mimc_pk.bin"), "wb").write(g.serialize_pk(g.equality_key(SETUP_SEED)))
open(os.path.join(out, "membership_mimc_pk.bin"), "wb").write(g.serialize_pk(g.membership_key(SETUP_SEED)))
mimc = {str(v): g.commit_value_snark(v).hex() for v in (0, 1, 42, 43, 99, 123, 2**32, 2**64 - 1)}
json.dump({"setup_seed": SETUP_SEED.hex(), "mimc": mimc}, open(os.path.join(out, "groth16_vectors.json"), "w"), indent=1)
print("ok")
