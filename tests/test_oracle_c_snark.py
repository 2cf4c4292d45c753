"""The C restatements of the SNARK and STARK provers (oracle/c/groth16.c, stark.c, batch.c) against the Python bigint
models, through committed vectors (tests/golden/snark_oracle_vectors.json, stark_oracle_vectors.json; generator scripts
beside them).  No GPU.  These ports exist so that bench.py's cpu_baseline can time the whole mixed batch."""
import ctypes
import hashlib
import json
import os

import numpy as np

from libzkp_amd import workloads as wl
from util import P, U64

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _keys(oracle_c):
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        pk = open(os.path.join(GOLD, name), "rb").read()
        assert oracle_c.zkp_oracle_g16_load_key(kind, pk, U64(len(pk))) == 0


def test_mimc_and_key_loading(oracle_c):
    vec = json.load(open(os.path.join(GOLD, "groth16_vectors.json")))
    out = ctypes.create_string_buffer(32)
    for v, want in vec["mimc"].items():
        assert oracle_c.zkp_oracle_snark_commit_value(U64(int(v)), out) == 0 and out.raw.hex() == want
    _keys(oracle_c)
    assert oracle_c.zkp_oracle_g16_load_key(0, b"\0" * 100, U64(100)) != 0        # malformed key
    pk = open(os.path.join(GOLD, "membership_mimc_pk.bin"), "rb").read()
    assert oracle_c.zkp_oracle_g16_load_key(0, pk, U64(len(pk))) != 0            # wrong circuit shape
    _keys(oracle_c)


def test_groth16_envelopes_equal_the_python_model(oracle_c):
    _keys(oracle_c)
    vec = json.load(open(os.path.join(GOLD, "snark_oracle_vectors.json")))
    out = ctypes.create_string_buffer(2048)
    ol = ctypes.c_uint32()
    for c in vec["equality"]:
        v = int(c["value"])
        assert oracle_c.zkp_oracle_prove_equality(U64(v), U64(v), bytes.fromhex(c["seed"]), out, 2048, ctypes.byref(ol)) == 0
        assert ol.value == 298 == c["len"] and out.raw[:298].hex() == c["envelope"]
    for c in vec["membership"]:
        s = [int(x) for x in c["set"]]
        arr = (ctypes.c_uint64 * len(s))(*s)
        assert oracle_c.zkp_oracle_prove_membership(U64(int(c["value"])), arr, len(s), bytes.fromhex(c["seed"]), out, 2048, ctypes.byref(ol)) == 0
        assert ol.value == c["len"] and out.raw[:ol.value].hex() == c["envelope"]
    # validation.rs:21-27,50-63
    assert oracle_c.zkp_oracle_prove_equality(U64(1), U64(2), bytes(32), out, 2048, ctypes.byref(ol)) == 1 and ol.value == 0
    arr = (ctypes.c_uint64 * 2)(5, 6)
    assert oracle_c.zkp_oracle_prove_membership(U64(7), arr, 2, bytes(32), out, 2048, ctypes.byref(ol)) == 1
    assert oracle_c.zkp_oracle_prove_membership(U64(5), arr, 0, bytes(32), out, 2048, ctypes.byref(ol)) == 1


def test_stark_envelopes_equal_the_python_model(oracle_c):
    vec = json.load(open(os.path.join(GOLD, "stark_oracle_vectors.json")))
    out = ctypes.create_string_buffer(4096)
    ol = ctypes.c_uint32()
    for c in vec["vectors"]:
        assert oracle_c.zkp_oracle_prove_improvement(U64(int(c["old"])), U64(int(c["new"])), out, 4096, ctypes.byref(ol)) == 0
        env = out.raw[:ol.value]
        assert len(env) == c["len"] and hashlib.sha256(env).hexdigest() == c["sha256"] and env[:64].hex() == c["head"] and env[-32:].hex() == c["tail"]
    assert oracle_c.zkp_oracle_prove_improvement(U64(5), U64(5), out, 4096, ctypes.byref(ol)) == 1 and ol.value == 0
    # live against the Python model on a few more inputs (the STARK is cheap in bigints)
    from oracle.py import stark
    rng = np.random.default_rng(17)
    for _ in range(6):
        old = int(rng.integers(0, 2**63)); new = old + 1 + int(rng.integers(0, 2**32))
        assert oracle_c.zkp_oracle_prove_improvement(U64(old), U64(new), out, 4096, ctypes.byref(ol)) == 0
        assert out.raw[:ol.value] == stark.prove_improvement(old, new)


def test_process_batch_port_matches_the_single_op_entries(oracle_c):
    _keys(oracle_c)
    ops, lists, seeds = wl.mixed_ops(16, 5)
    ops = ops.copy()
    ops["a"][1] += np.uint64(1)                              # one invalid equality op: fails the batch, leaves no bytes
    cap = wl.max_output_bytes(ops)
    out = np.zeros(cap, dtype=np.uint8); off = np.zeros(17, dtype=np.uint64); st = np.zeros(16, dtype=np.int32)
    rc = oracle_c.zkp_oracle_process_batch(U64(16), P(ops), P(lists), P(seeds), P(out), U64(cap), P(off), P(st), 4)
    assert rc == 1 and st[1] == 1 and off[1] == off[2] and not st[[0] + list(range(2, 16))].any()
    one = ctypes.create_string_buffer(4096); ol = ctypes.c_uint32()
    sd = seeds.reshape(16, 32)
    for i in range(16):
        o = ops[i]; k = int(o["kind"])
        if i == 1:
            continue
        if k == wl.OP_RANGE:
            assert oracle_c.zkp_oracle_prove_range(U64(int(o["a"])), U64(int(o["b"])), U64(int(o["c"])), 64, sd[i].tobytes(), one, 4096, ctypes.byref(ol)) == 0
        elif k == wl.OP_EQUALITY:
            assert oracle_c.zkp_oracle_prove_equality(U64(int(o["a"])), U64(int(o["b"])), sd[i].tobytes(), one, 4096, ctypes.byref(ol)) == 0
        elif k == wl.OP_MEMBERSHIP:
            s = lists[int(o["list_off"]):int(o["list_off"]) + int(o["count"])].copy()
            assert oracle_c.zkp_oracle_prove_membership(U64(int(o["a"])), P(s), int(o["count"]), sd[i].tobytes(), one, 4096, ctypes.byref(ol)) == 0
        else:
            assert oracle_c.zkp_oracle_prove_improvement(U64(int(o["a"])), U64(int(o["b"])), one, 4096, ctypes.byref(ol)) == 0
        assert out[int(off[i]):int(off[i + 1])].tobytes() == one.raw[:ol.value]
    small = np.zeros(10, dtype=np.uint8)
    assert oracle_c.zkp_oracle_process_batch(U64(16), P(ops), P(lists), P(seeds), P(small), U64(10), P(off), P(st), 4) == 100
