"""The range-proof verifier steps (libzkp_amd/csrc/bp_verify.h) run on the host against the oracle's verifier (no GPU):
same accept / reject verdicts on valid proofs, every kind of tampering the reference tests use, and structural garbage."""
import ctypes
import os
import random

import numpy as np
import pytest

from util import P, U64, oracle_prove, oracle_verify, outputs, workload


@pytest.fixture(scope="module")
def emul():
    import __graft_entry__ as ge
    ge.build_emul()
    L = ctypes.CDLL(os.path.join(ge.EMUL_DIR, "_build", "libemul_bp.so"))
    L.emul_verify_range_batch.argtypes = [ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    return L


def emul_verify(L, out, lens, mn, mx, nchunks=5):
    n = out.shape[0]
    ok = np.zeros(n, dtype=np.uint8)
    assert L.emul_verify_range_batch(n, P(out), out.shape[1], P(lens), P(mn), P(mx), P(ok), nchunks) == 0
    return ok


def test_scalarmult_matches_libsodium_fixture(emul, golden_ristretto):
    out = (ctypes.c_uint32 * 8)()
    W = lambda b: (ctypes.c_uint32 * 8)(*[int.from_bytes(b[4 * i:4 * i + 4], "little") for i in range(8)])  # noqa: E731
    n = 0
    for c in golden_ristretto["from_hash_add_sub_mul"][:12]:
        p, k, want = bytes.fromhex(c["p"]), bytes.fromhex(c["k"]), bytes.fromhex(c["k_times_p"])
        kk = int.from_bytes(k, "little")
        if kk >= 2**253:
            continue
        assert emul.emul_scalarmult(W(p), W(k), out) == 1
        assert b"".join(int(x).to_bytes(4, "little") for x in out) == want
        n += 1
    assert n >= 4
    seen = set()
    for c in golden_ristretto["validity"]:          # libsodium's verdict on random / edge encodings (it ignores bit 255: masked in the fixture)
        assert emul.emul_scalarmult(W(bytes.fromhex(c["bytes"])), W(bytes(32)), out) == (1 if c["valid"] else 0)
        seen.add(c["valid"])
    assert seen == {True, False}


def test_verdicts_equal_oracle(emul, oracle_c):
    rnd = random.Random(31)
    n = 6
    v, mn, mx, seeds = workload(n, 77)
    v[0], v[1] = 0, 2**32                                  # both bounds tight
    rc, out, lens, st = oracle_prove(oracle_c, v, mn, mx, seeds)
    assert rc == 0
    cases = [out.copy()]
    # tampering: one flipped bit in each region of the envelope of proof 0..n-1 (header, bounds, both sub-proofs, commitments)
    for pos in (0, 1, 3, 12, 20, 31, 40, 100, 200, 300, 500, 700, 716, 800, 1000, 1300, 1390, 1420, 1446, 1477):
        t = out.copy()
        for i in range(n):
            t[i, (pos + 37 * i) % 1478] ^= 1 << rnd.randrange(8)
        cases.append(t)
    for t in cases:
        want = oracle_verify(oracle_c, t, lens, mn, mx)[1]
        got = emul_verify(emul, t, lens, mn, mx)
        assert list(got) == list(want)
    assert list(emul_verify(emul, out, lens, mn, mx, nchunks=1)) == [1] * n      # chunking does not matter
    # wrong bounds (bulletproofs.rs:704), wrong length, truncated, min > max
    assert list(emul_verify(emul, out, lens, mn + 1, mx)) == [0] * n
    assert list(emul_verify(emul, out, lens, mn, mx - 1)) == [0] * n
    short = lens.copy(); short[:] = 1477
    assert list(emul_verify(emul, out, short, mn, mx)) == [0] * n
    assert list(emul_verify(emul, out, lens, mx, mn)) == [0] * n
    # proofs swapped between ops with different values: commitment binding
    sw = out.copy(); sw[0, 30:1382], sw[2, 30:1382] = out[2, 30:1382], out[0, 30:1382]
    want = oracle_verify(oracle_c, sw, lens, mn, mx)[1]
    assert list(emul_verify(emul, sw, lens, mn, mx)) == list(want) and want[0] == 0 and want[2] == 0


def test_verdicts_bit_widths(emul, oracle_c):
    """Proofs made with prove_range_with_bits (8/16/32-bit generators; the width travels in the envelope,
    bulletproofs.rs:163-164,211-216) mixed in one verification batch with 64-bit ones."""
    rnd = random.Random(5)
    rows, lens_all, mns, mxs = [], [], [], []
    for bits in (8, 16, 32, 64):
        cap = 2**bits - 1
        v = np.array([cap, 3], dtype=np.uint64); mn = np.array([0, 1], dtype=np.uint64); mx = np.array([cap, 9], dtype=np.uint64)
        seeds = np.frombuffer(np.random.default_rng(100 + bits).bytes(64), dtype=np.uint8).copy()
        out, lens, st = outputs(2)
        assert oracle_c.zkp_oracle_prove_range_batch(U64(2), P(v), P(mn), P(mx), bits, P(seeds), P(out), U64(1478), P(lens), P(st), 2) == 0
        rows.append(out); lens_all.append(lens); mns.append(mn); mxs.append(mx)
    out, lens, mn, mx = np.concatenate(rows), np.concatenate(lens_all), np.concatenate(mns), np.concatenate(mxs)
    n = len(lens)
    assert list(lens) == [1094, 1094, 1222, 1222, 1350, 1350, 1478, 1478]
    assert list(emul_verify(emul, out, lens, mn, mx)) == [1] * n
    cases = []
    for pos in (26, 30, 40, 300, 500, 700, 1000):          # the width field, a length field, proof bytes
        t = out.copy()
        for i in range(n):
            t[i, (pos + 13 * i) % int(lens[i])] ^= 1 << rnd.randrange(8)
        cases.append(t)
    t = out.copy(); t[0, 26] = 16; cases.append(t)         # claims another valid width than the proof was made for
    for t in cases:
        want = oracle_verify(oracle_c, t, lens, mn, mx)[1]
        assert list(emul_verify(emul, t, lens, mn, mx)) == list(want)
    assert sum(int(x) for t in cases for x in oracle_verify(oracle_c, t, lens, mn, mx)[1]) < len(cases) * n
