"""C oracle == Python model on the committed vectors, plus its verifier on the reference's negative cases (no GPU)."""
import ctypes

import numpy as np

from util import P, U64, oracle_prove, oracle_verify, workload


def test_generators_and_tape(oracle_c, golden_bp):
    enc = ctypes.create_string_buffer(32)
    for i in range(130):
        oracle_c.zkp_oracle_generator(i, enc)
        assert enc.raw.hex() == golden_bp["generators"][str(i)]
    out = ctypes.create_string_buffer(64)
    for t in golden_bp["tape"]:
        oracle_c.zkp_oracle_tape_draw64(bytes.fromhex(t["seed"]), t["proof_idx"], t["slot"], out)
        assert out.raw.hex() == t["draw64"]


def test_single_vectors(oracle_c, golden_bp):
    for c in golden_bp["single"]:
        pr = ctypes.create_string_buffer(len(c["proof"]) // 2)
        V = ctypes.create_string_buffer(32)
        rc = oracle_c.zkp_oracle_prove_single(c["label"].encode(), U64(c["v"]), bytes.fromhex(c["blinding"]), c["n_bits"],
                                              bytes.fromhex(c["seed"]), c["proof_idx"], pr, V)
        assert rc == 0 and pr.raw.hex() == c["proof"] and V.raw.hex() == c["commitment"]
        assert oracle_c.zkp_oracle_verify_single(c["label"].encode(), pr.raw, len(pr.raw), V.raw, c["n_bits"]) == 1
        assert oracle_c.zkp_oracle_verify_single(b"other", pr.raw, len(pr.raw), V.raw, c["n_bits"]) == 0


def test_range_threshold_consistency_vectors(oracle_c, golden_bp):
    out = ctypes.create_string_buffer(8192)
    ol = ctypes.c_uint32()
    for c in golden_bp["range"]:
        rc = oracle_c.zkp_oracle_prove_range(U64(c["value"]), U64(c["min"]), U64(c["max"]), 64, bytes.fromhex(c["seed"]), out, 8192, ctypes.byref(ol))
        assert rc == 0 and out.raw[: ol.value].hex() == c["proof"]
        env = bytes.fromhex(c["proof"])
        assert oracle_c.zkp_oracle_verify_range(env, len(env), U64(c["min"]), U64(c["max"])) == 1
        if c["max"] > c["min"]:
            assert oracle_c.zkp_oracle_verify_range(env, len(env), U64(c["min"]), U64(c["max"] - 1)) == 0
        bad = bytearray(env)
        bad[12] ^= 1
        assert oracle_c.zkp_oracle_verify_range(bytes(bad), len(bad), U64(c["min"]), U64(c["max"])) == 0
    for c in golden_bp["threshold"]:
        vals = (ctypes.c_uint64 * len(c["values"]))(*c["values"])
        rc = oracle_c.zkp_oracle_prove_threshold(vals, len(c["values"]), U64(c["threshold"]), 64, bytes.fromhex(c["seed"]), out, 8192, ctypes.byref(ol))
        assert rc == 0 and ol.value == 762 and out.raw[: ol.value].hex() == c["proof"]
        env = bytes.fromhex(c["proof"])
        assert oracle_c.zkp_oracle_verify_threshold(env, len(env), U64(c["threshold"])) == 1
        assert oracle_c.zkp_oracle_verify_threshold(env, len(env), U64(c["threshold"] + 1)) == 0
    for c in golden_bp["consistency"]:
        d = (ctypes.c_uint64 * len(c["data"]))(*c["data"])
        rc = oracle_c.zkp_oracle_prove_consistency(d, len(c["data"]), bytes.fromhex(c["seed"]), out, 8192, ctypes.byref(ol))
        assert rc == 0 and out.raw[: ol.value].hex() == c["proof"]
        env = bytes.fromhex(c["proof"])
        assert oracle_c.zkp_oracle_verify_consistency(env, len(env)) == 1


def test_validation_status_codes(oracle_c):
    out = ctypes.create_string_buffer(2048)
    ol = ctypes.c_uint32()
    seed = bytes(32)
    assert oracle_c.zkp_oracle_prove_range(U64(11), U64(0), U64(10), 64, seed, out, 2048, ctypes.byref(ol)) == 1
    assert oracle_c.zkp_oracle_prove_range(U64(5), U64(10), U64(0), 64, seed, out, 2048, ctypes.byref(ol)) == 1
    assert oracle_c.zkp_oracle_prove_range(U64(5), U64(0), U64(10), 64, seed, out, 100, ctypes.byref(ol)) == 100


def test_batch_prove_verify_roundtrip(oracle_c):
    v, mn, mx, seeds = workload(12, 3)
    rc, out, lens, st = oracle_prove(oracle_c, v, mn, mx, seeds, threads=4)
    assert rc == 0 and (lens == 1478).all()
    allok, ok = oracle_verify(oracle_c, out, lens, mn, mx, threads=4)
    assert allok == 1
    out[5, 700] ^= 0x10
    allok, ok = oracle_verify(oracle_c, out, lens, mn, mx, threads=4)
    assert allok == 0 and ok.sum() == 11 and ok[5] == 0
