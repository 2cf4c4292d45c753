"""Device math headers and per-thread prover steps, compiled for the host (tests/emul), against the oracle (no GPU)."""
import ctypes
import hashlib
import random

import numpy as np

from oracle.py import bulletproofs as bp
from oracle.py import merlin, ristretto as R
from util import P, U64, oracle_prove, oracle_verify, outputs, workload

PP, LL = R.P, R.L


def W(x, n=8):
    return (ctypes.c_uint32 * n)(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)])


def I(w):
    return sum(int(w[i]) << (32 * i) for i in range(len(w)))


def test_field(emul):
    lib, _ = emul
    rnd = random.Random(1)
    out = (ctypes.c_uint32 * 8)()
    vals = [0, 1, 2, 19, PP - 1, PP - 2, PP - 19, 2**255 - 20, 2**254, 2**26 - 1] + [rnd.randrange(PP) for _ in range(200)]
    for i, a in enumerate(vals):
        b, c = vals[(i * 7 + 3) % len(vals)], vals[(i * 5 + 1) % len(vals)]
        for op, f in ((0, a * b), (1, a * a), (2, a + b), (3, a - b), (4, -a)):
            lib.emul_fe_op(op, W(a), W(b), out)
            assert I(out) == f % PP
        lib.emul_fe_loose(W(a), W(b), W(c), out)
        assert I(out) == (2 * a - c) * (b - c) % PP
        lib.emul_fe_op(6, W(a), W(b), out)
        assert I(out) == R._abs(a)
    for a in vals[:30]:
        lib.emul_fe_op(5, W(a), W(0), out)
        assert I(out) == pow(a, (PP - 5) // 8, PP)
        v = vals[(7 * a) % len(vals)]
        sq = lib.emul_sqrt_ratio(W(a), W(v), out)
        assert (bool(sq), I(out)) == R.sqrt_ratio_m1(a, v)


def test_scalars(emul):
    lib, _ = emul
    rnd = random.Random(2)
    out = (ctypes.c_uint32 * 8)()
    vals = [0, 1, 2, LL - 1, LL - 2, 2**252, 2**252 - 1] + [rnd.randrange(LL) for _ in range(150)]
    for i, a in enumerate(vals):
        b = vals[(i * 3 + 1) % len(vals)]
        for op, f in ((0, a * b), (1, a + b), (2, a - b), (4, -a)):
            lib.emul_sc_op(op, W(a), W(b), out)
            assert I(out) == f % LL
        if a:                                   # safegcd inversion (product path)
            lib.emul_sc_op(3, W(a), W(b), out)
            assert I(out) == pow(a, LL - 2, LL)
        if a and i < 12:                        # Fermat ladder cross-check
            lib.emul_sc_op(5, W(a), W(b), out)
            assert I(out) == pow(a, LL - 2, LL)
        dg10 = (ctypes.c_int16 * 26)()
        lib.emul_sc_recode1024(W(a), dg10)
        assert sum(int(dg10[k]) << (10 * k) for k in range(26)) == a and all(-511 <= int(x) <= 512 for x in dg10)
    for x in [0, 2**512 - 1, 2**256, LL * LL] + [rnd.randrange(2**512) for _ in range(50)]:
        lib.emul_sc_from_wide(W(x, 16), out)
        assert I(out) == x % LL
    for x in (2**256 - 1, LL, LL + 1, 2**255):   # from_bytes_mod_order on unreduced input
        lib.emul_sc_op(1, W(x), W(0), out)
        assert I(out) == x % LL


def test_group_tape_merlin(emul, golden_ristretto):
    lib, _ = emul
    out = (ctypes.c_uint32 * 8)()
    for c in golden_ristretto["from_hash_add_sub_mul"][:10]:
        h1, h2 = bytes.fromhex(c["hash_p"]), bytes.fromhex(c["hash_q"])
        lib.emul_from_uniform_encode(W(int.from_bytes(h1, "little"), 16), out)
        assert I(out).to_bytes(32, "little").hex() == c["p"]
        lib.emul_ge_lincomb(W(int.from_bytes(h1, "little"), 16), W(int.from_bytes(h2, "little"), 16), 1, 1, out)
        assert I(out).to_bytes(32, "little").hex() == c["p_plus_q"]
        a, b = 0xDEADBEEF, 12345
        lib.emul_ge_lincomb(W(int.from_bytes(h1, "little"), 16), W(int.from_bytes(h2, "little"), 16), a, b, out)
        assert I(out).to_bytes(32, "little") == (a * R.from_uniform_bytes(h1) + b * R.from_uniform_bytes(h2)).encode()
    lib.emul_ge_lincomb(W(1, 16), W(2, 16), 0, 0, out)
    assert I(out) == 0   # identity encodes to zeros
    o16 = (ctypes.c_uint32 * 16)()
    for t in range(8):
        seed = hashlib.sha256(b"s%d" % t).digest()
        lib.emul_tape_draw64(W(int.from_bytes(seed, "little")), 77 * t, 3 * t, o16)
        assert I(o16).to_bytes(64, "little") == bp.draw64(seed, 77 * t, 3 * t)
    o8 = (ctypes.c_uint32 * 8)()
    lib.emul_merlin_kat(b"test protocol", b"some label", b"some data", b"challenge", 8, o8)
    assert I(o8).to_bytes(32, "little").hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"
    rnd = random.Random(5)
    for t in range(6):   # long messages cross the 166-byte STROBE rate
        lab, m1 = b"L" * rnd.randrange(1, 30), bytes(rnd.randrange(256) for _ in range(rnd.randrange(0, 400))).hex().encode()
        o = (ctypes.c_uint32 * 60)()
        lib.emul_merlin_kat(lab, b"l1", m1, b"ch", 60, o)
        tr = merlin.Transcript(lab)
        tr.append_message(b"l1", m1)
        assert I(o).to_bytes(240, "little") == tr.challenge_bytes(b"ch", 240)


def test_prover_steps_equal_oracle(emul, oracle_c, golden_bp):
    """The restructured (fixed-generator, fold-the-coefficients) kernels give the oracle's bytes."""
    _, lib = emul
    enc = (ctypes.c_uint32 * 8)()
    for i in (0, 1, 2, 65, 66, 129):
        lib.emul_generator(i, enc)
        assert bytes(enc).hex() == golden_bp["generators"][str(i)]
    n = 5
    v, mn, mx, seeds = workload(n, 11)
    v[1], v[2] = 0, 2**32
    mn[3], mx[3], v[3] = 5, 2**64 - 1, 2**63 + 12345
    mn[4], mx[4], v[4] = 9, 9, 9
    for budget in (128, 32, 1000, 10000 + 7, 10000 + 128):      # slot-aligned budgets and window-granular even layouts
        out, lens, st = outputs(n)
        rc = lib.emul_prove_range_batch(U64(n), P(v), P(mn), P(mx), P(seeds), P(out), U64(1478), P(lens), P(st), budget)
        rc2, o2, l2, s2 = oracle_prove(oracle_c, v, mn, mx, seeds, threads=4)
        assert rc == 0 and rc2 == 0 and (lens == 1478).all()
        assert (out == o2).all()
    # golden vector through the emulated kernels
    c = golden_bp["range"][0]
    g = lambda x: np.array([x], dtype=np.uint64)  # noqa: E731
    out, lens, st = outputs(1)
    sd = np.frombuffer(bytes.fromhex(c["seed"]), dtype=np.uint8).copy()
    lib.emul_prove_range_batch(U64(1), P(g(c["value"])), P(g(c["min"])), P(g(c["max"])), P(sd), P(out), U64(1478), P(lens), P(st), 128)
    assert out[0].tobytes().hex() == c["proof"]


def test_steps_validation(emul):
    _, lib = emul
    n = 3
    v = np.array([5, 11, 5], dtype=np.uint64)
    mn = np.array([0, 0, 10], dtype=np.uint64)
    mx = np.array([10, 10, 0], dtype=np.uint64)
    seeds = np.zeros(32 * n, dtype=np.uint8)
    out, lens, st = outputs(n)
    rc = lib.emul_prove_range_batch(U64(n), P(v), P(mn), P(mx), P(seeds), P(out), U64(1478), P(lens), P(st), 128)
    assert rc == 1 and list(st) == [0, 1, 1] and list(lens) == [1478, 0, 0]


def test_prover_steps_bit_widths(emul, oracle_c):
    """prove_range_with_bits (range_proof.rs:16-27, bulletproofs.rs:112-178): 8/16/32-bit proofs through the same emulated
    kernels give the oracle's bytes (C restatement, cross-checked against the Python one for one case)."""
    import oracle.py.bulletproofs as obp
    _, lib = emul
    for bits in (8, 16, 32):
        size = 1478 - 2 * 64 * (6 - bits.bit_length() + 1)
        cap = 2**bits - 1
        v = np.array([cap, 5, 0, 7, 300 if bits == 8 else 3], dtype=np.uint64)
        mn = np.array([0, 5, 0, 3, 0], dtype=np.uint64)
        mx = np.array([cap, 5 + cap, cap, 7, 300 if bits == 8 else 9], dtype=np.uint64)
        if bits == 8:
            mx[4] = 310                                              # value - min = 300 needs 9 bits: "range width exceeds 8-bit capacity"
        n = len(v)
        seeds = np.frombuffer(np.random.default_rng(bits).bytes(32 * n), dtype=np.uint8).copy()
        out, lens, st = outputs(n)
        rc = lib.emul_prove_range_batch_bits(U64(n), P(v), P(mn), P(mx), bits, P(seeds), P(out), U64(1478), P(lens), P(st), 10000 + 9)
        o2, l2, s2 = outputs(n)
        rc2 = oracle_c.zkp_oracle_prove_range_batch(U64(n), P(v), P(mn), P(mx), bits, P(seeds), P(o2), U64(1478), P(l2), P(s2), 4)
        want_fail = bits == 8
        assert rc == rc2 == (1 if want_fail else 0)
        assert list(lens) == list(l2) == [size] * 4 + [0 if want_fail else size]
        assert list(st != 0) == list(s2 != 0)
        assert (out[st == 0] == o2[st == 0]).all()                   # (the C ABI wrapper clears the bytes of failed items)
        allok, ok = oracle_verify(oracle_c, out, lens, mn, mx, threads=4)
        assert list(ok) == [1] * 4 + [0 if want_fail else 1]
        if bits == 8:
            body, commit = obp._unwire(obp.prove_range_with_bounds_bits(int(v[0]), int(mn[0]), int(mx[0]), 8, seeds[:32].tobytes()))
            env = bytes([2, 1]) + len(body).to_bytes(4, "little") + (32).to_bytes(4, "little") + body + commit   # proof/mod.rs:23-36
            assert env == out[0, :size].tobytes()
    out, lens, st = outputs(1)
    one = np.array([1], dtype=np.uint64)
    assert lib.emul_prove_range_batch_bits(U64(1), P(one), P(one), P(one), 12, P(np.zeros(32, dtype=np.uint8)), P(out), U64(1478), P(lens), P(st), 128) == -2


def test_hbm_table_builder_and_radix_65536_digits(emul):
    """The prover's HBM-resident generator tables (libzkp_amd/csrc/edg.h, round 4): the signed radix-2^16 recoding reassembles to the
    scalar; the device builder's steps produce, for whole windows, the multiples double-and-add gives; its self-check passes on a good
    window and finds a flipped bit; one fixed-base term assembled from digits and entries equals k * G."""
    _, lib = emul
    L = 2**252 + 27742317777372353535851937790883648493
    rng = np.random.default_rng(11)
    cases = [0, 1, 32767, 32768, 65535, 65536, 2**64 - 1, 2**253 - 1, L - 1] + [int.from_bytes(rng.bytes(32), "little") % L for _ in range(40)]
    for k in cases:
        raw = np.frombuffer(k.to_bytes(32, "little"), dtype=np.uint32).copy()
        out = np.zeros(8, dtype=np.uint32)
        lib.emul_sc_recode65536(P(raw), P(out))
        digs = out.view(np.int16).astype(object)
        assert sum(int(d) << (16 * i) for i, d in enumerate(digs)) == k
        assert all(-32768 <= int(d) <= 32767 for d in digs)
        if k < 2**64:
            assert not any(int(d) for d in digs[5:])          # a 64-bit value stays inside five windows
    for gen, w in ((0, 0), (1, 15), (66, 7), (129, 3)):
        assert lib.emul_edg_window(gen, w, 12, 5 + w) == 0
    enc = np.zeros(8, dtype=np.uint32)
    for gen, k in ((0, 1), (2, L - 1), (1, cases[12]), (70, 2**252 + 5), (129, 32768 * (2**16) ** 3)):
        raw = np.frombuffer(int(k).to_bytes(32, "little"), dtype=np.uint32).copy()
        assert lib.emul_edg_term(gen, P(raw), P(enc)) == 1
