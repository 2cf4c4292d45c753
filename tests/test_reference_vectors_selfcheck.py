"""tests/test_reference_vectors.py is skipped in this repository (its vectors can only come from the real libzkp).  So that the
consumer is not dead code the day vectors arrive, this file feeds it STAND-IN vectors in the same JSON shapes -- made by oracle/c, i.e.
pinning nothing -- through the ZKP_REFERENCE_VECTORS override and expects every case to run and pass."""
import ctypes
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from libzkp_amd import workloads as wl
from util import P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
U64 = ctypes.c_uint64


def _stand_in_vectors(orc, d):
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        shutil.copy(os.path.join(GOLD, name), os.path.join(d, name))
        blob = open(os.path.join(GOLD, name), "rb").read()
        assert orc.zkp_oracle_g16_load_key(kind, blob, U64(len(blob))) == 0

    def prove(kind, count, a, b, c, lists, seed):
        op = np.zeros(1, dtype=wl.OP_DTYPE)
        op["kind"], op["count"], op["a"], op["b"], op["c"] = kind, count, a, b, c
        ls = np.array(lists or [0], dtype=np.uint64)
        sd = np.frombuffer(seed, dtype=np.uint8).copy()
        cap = wl.max_output_bytes(op)
        out = np.zeros(cap, dtype=np.uint8); off = np.zeros(2, dtype=np.uint64); st = np.zeros(1, dtype=np.int32)
        assert orc.zkp_oracle_process_batch(U64(1), P(op), P(ls), P(sd), P(out), U64(cap), P(off), P(st), 1) == 0 and st[0] == 0
        return out[:int(off[1])].tobytes().hex()

    sd = lambda i: bytes([i]) * 32  # noqa: E731
    hips = [dict(scheme="range", value=7, min=0, max=10, seed=sd(1).hex(), envelope=prove(1, 0, 7, 0, 10, None, sd(1))),
            dict(scheme="threshold", values=[3, 4, 5], threshold=10, seed=sd(2).hex(), envelope=prove(3, 3, 10, 0, 0, [3, 4, 5], sd(2))),
            dict(scheme="consistency", values=[1, 2, 3], seed=sd(3).hex(), envelope=prove(6, 3, 0, 0, 0, [1, 2, 3], sd(3))),
            dict(scheme="equality", value=42, seed=sd(4).hex(), envelope=prove(2, 0, 42, 42, 0, None, sd(4))),
            dict(scheme="membership", value=2, set=[1, 2, 3], seed=sd(5).hex(), envelope=prove(4, 3, 2, 0, 0, [1, 2, 3], sd(5)))]
    json.dump(hips, open(os.path.join(d, "hip_envelopes.json"), "w"))
    json.dump([dict(old=o, new=n, envelope=prove(5, 0, o, n, 0, None, sd(0))) for o, n in ((1, 5), (30, 50))], open(os.path.join(d, "improvement_vectors.json"), "w"))
    cms = []
    for v in (0, 42, 2**64 - 1):
        out = (ctypes.c_uint8 * 32)()
        assert orc.zkp_oracle_snark_commit_value(U64(v), out) == 0
        cms.append(dict(value=v, commitment=bytes(out).hex()))
    json.dump(cms, open(os.path.join(d, "snark_commitments.json"), "w"))
    refs = [dict(scheme="range", verify_args=dict(min=0, max=10), envelope=hips[0]["envelope"]),
            dict(scheme="threshold", verify_args=dict(threshold=10), envelope=hips[1]["envelope"]),
            dict(scheme="consistency", verify_args={}, envelope=hips[2]["envelope"]),
            dict(scheme="equality", verify_args=dict(value=42), envelope=hips[3]["envelope"]),
            dict(scheme="membership", verify_args=dict(set=[1, 2, 3]), envelope=hips[4]["envelope"]),
            dict(scheme="improvement", verify_args=dict(old=1), envelope=prove(5, 0, 1, 5, 0, None, sd(0)))]
    json.dump(refs, open(os.path.join(d, "reference_envelopes.json"), "w"))
    # crafted points at infinity in the equality envelope (A, B, C in turn), verdicts by the oracle's pairing verifier
    from oracle.py import groth16 as g
    vk = g.vk_from_pk_bytes(open(os.path.join(GOLD, "equality_mimc_pk.bin"), "rb").read())
    specials = []
    base = bytes.fromhex(hips[3]["envelope"])
    assert g.verify_equality_envelope_under(vk, base)
    for what, at, ln in (("A", 10, 64), ("B", 74, 128), ("C", 202, 64)):
        e = bytearray(base); e[at:at + ln] = bytes(ln); e[at + ln - 1] = 0x40
        specials.append(dict(scheme="equality", what=what + " at infinity", verify_args=dict(value=42), envelope=bytes(e).hex(),
                             reference_verdict=bool(g.verify_equality_envelope_under(vk, bytes(e)))))
    json.dump(specials, open(os.path.join(d, "special_envelopes.json"), "w"))


def _run_consumer(d, marker):
    env = dict(os.environ, ZKP_REFERENCE_VECTORS=d)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_reference_vectors.py"), "-q", "-m", marker, "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    return p.returncode, p.stdout[-1500:] + p.stderr[-500:]


def test_consumer_runs_every_cpu_case_on_stand_in_vectors(oracle_c, tmp_path):
    _stand_in_vectors(oracle_c, str(tmp_path))
    rc, tail = _run_consumer(str(tmp_path), "not gpu")
    assert rc == 0 and "4 passed" in tail and "skipped" not in tail, tail


@pytest.mark.gpu
def test_consumer_runs_every_gpu_case_on_stand_in_vectors(oracle_c, tmp_path):
    _stand_in_vectors(oracle_c, str(tmp_path))
    rc, tail = _run_consumer(str(tmp_path), "gpu")
    assert rc == 0 and "4 passed" in tail and "skipped" not in tail, tail
