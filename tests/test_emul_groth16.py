"""BN254 device headers and the Groth16 witness/QAP steps compiled for the host, against the oracle (no GPU)."""
import ctypes
import os
import random

import numpy as np
import pytest

from oracle.py import bn254 as b
from oracle.py import groth16 as g

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libs():
    import __graft_entry__ as ge
    ge.build_emul()
    d = os.path.join(ge.EMUL_DIR, "_build")
    return ctypes.CDLL(os.path.join(d, "libemul_bn254.so")), ctypes.CDLL(os.path.join(d, "libemul_g16.so"))


def W(x, n=8):
    return (ctypes.c_uint32 * n)(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)])


def I(w):
    return sum(int(w[i]) << (32 * i) for i in range(len(w)))


def test_fields(libs):
    lib, _ = libs
    rnd = random.Random(7)
    out = (ctypes.c_uint32 * 8)()
    for fn, m in ((lib.emul_fq_op, b.P), (lib.emul_fr_op, b.R)):
        vals = [0, 1, 2, m - 1, m - 2, (m - 1) // 2, (m + 1) // 2, 2**253, 2**255, 2**256 - 1] + [rnd.randrange(m) for _ in range(150)]
        for i, a in enumerate(vals):
            c = vals[(i * 7 + 3) % len(vals)]
            for op, f in ((0, a * c), (1, a + c), (2, a - c), (3, -a), (5, (2 * c) * (a - 2 * c))):
                fn(op, W(a), W(c), out)
                assert I(out) == f % m
            if a % m and i < 15:
                fn(4, W(a), W(c), out)
                assert I(out) == pow(a, -1, m)
    # the nine-limb form of the G1 MSM loop (bn254_fq9.h) through its conversions: product, squaring, fused double product, subtractions
    o2, o3 = (ctypes.c_uint32 * 8)(), (ctypes.c_uint32 * 8)()
    vals = [0, 1, 2, b.P - 1, b.P - 2, (b.P - 1) // 2, 2**253, 2**255, 2**256 - 1] + [rnd.randrange(b.P) for _ in range(150)]
    for i, a in enumerate(vals):
        c = vals[(i * 7 + 3) % len(vals)]
        lib.emul_fq9_mul(W(a), W(c), out)
        assert I(out) == a * c % b.P
        lib.emul_fq9_ops(W(a), W(c), out, o2, o3)
        assert I(out) == a * a % b.P and I(o2) == (a * a - a * c) % b.P and I(o3) == 3 * (a - c) % b.P
    vals = [0, 1, 2, b.R - 1, b.R - 2, (b.R - 1) // 2, 2**253, 2**255, 2**256 - 1] + [rnd.randrange(b.R) for _ in range(150)]
    for i, a in enumerate(vals):                       # Fr on nine limbs (bn254_fr9.h, the QAP step's form)
        c = vals[(i * 7 + 3) % len(vals)]
        lib.emul_fr9_ops(W(a), W(c), out, o2, o3)
        assert I(out) == a * c % b.R and I(o2) == (a - c) % b.R and I(o3) == (a + 20 * c) % b.R
    for x in [0, 2**512 - 1] + [rnd.randrange(2**512) for _ in range(20)]:
        lib.emul_fr_from_wide(W(x, 16), out)
        assert I(out) == x % b.R


def test_groups_and_serialisation(libs):
    lib, _ = libs
    rnd = random.Random(9)
    g1w = lambda pt: W(pt[0] | (pt[1] << 256), 16)  # noqa: E731
    g2w = lambda pt: W(pt[0][0] | (pt[0][1] << 256) | (pt[1][0] << 512) | (pt[1][1] << 768), 32)  # noqa: E731
    o16, o32 = (ctypes.c_uint32 * 16)(), (ctypes.c_uint32 * 32)()
    for _ in range(3):
        k1, k2, a, c = (rnd.randrange(1, b.R) for _ in range(4))
        p1, q1 = b.G1C.mul_pt(b.G1, a), b.G1C.mul_pt(b.G1, c)
        lib.emul_g1_lincomb(g1w(p1), g1w(q1), W(k1), W(k2), o16)
        assert I(o16).to_bytes(64, "little") == b.ser_g1(b.G1C.add_pts(b.G1C.mul_pt(p1, k1), b.G1C.mul_pt(q1, k2)))
        p2, q2 = b.G2C.mul_pt(b.G2, a), b.G2C.mul_pt(b.G2, c)
        lib.emul_g2_lincomb(g2w(p2), g2w(q2), W(k1), W(k2), o32)
        assert I(o32).to_bytes(128, "little") == b.ser_g2(b.G2C.add_pts(b.G2C.mul_pt(p2, k1), b.G2C.mul_pt(q2, k2)))
    lib.emul_g1_lincomb(g1w(b.G1), g1w(b.G1), W(5), W(b.R - 5), o16)
    assert I(o16).to_bytes(64, "little") == b.ser_g1(None)
    for signs in ([1], [1, 1], [1, -1], [1, 1, 1, -1, -1, -1], [1, 1, -1, 1, 1, 1], [-1, -1, -1]):   # exceptional cases of madd
        arr = (ctypes.c_int * len(signs))(*signs)
        s = sum(signs) % b.R
        lib.emul_g1_madd_chain(g1w(b.G1), arr, len(signs), o16)
        assert I(o16).to_bytes(64, "little") == b.ser_g1(b.G1C.mul_pt(b.G1, s) if s else None)
        lib.emul_g2_madd_chain(g2w(b.G2), arr, len(signs), o32)
        assert I(o32).to_bytes(128, "little") == b.ser_g2(b.G2C.mul_pt(b.G2, s) if s else None)


def test_fq_lazy_limbs(libs):
    """Unsaturated Fq: products and weak reductions on uncarried, loosely reduced inputs (bn254_fq.h bounds vocabulary)."""
    lib, _ = libs
    rnd = random.Random(11)
    out = (ctypes.c_uint32 * 8)()
    p = b.P
    edge = [0, 1, p - 1, p - 2, (p - 1) // 2, 2**253, 2**254 - 1]
    for trial in range(300):
        a = edge[trial % len(edge)] if trial < 3 * len(edge) else rnd.randrange(p)
        c = edge[(trial // len(edge)) % len(edge)] if trial < 3 * len(edge) else rnd.randrange(p)
        ka, kb = rnd.randrange(0, 7), rnd.randrange(0, 7)
        for op, f in ((0, a * c), (1, (a + c) * (a - c)), (2, a - c), (3, a), (4, (a - 4 * c) ** 2)):
            lib.emul_fq_lazy(op, W(a), W(c), ka if op != 4 else min(ka, 3), kb if op != 2 else min(kb, 5), out)
            assert I(out) == f % p, (op, trial)


def test_msm_inner_loop_lazy(libs):
    """The unchecked, lazily reduced mixed additions of the MSM kernels, started from an offset point with Z != 1."""
    lib, _ = libs
    rnd = random.Random(13)
    g1w = lambda pt: [(pt[0] >> (32 * i)) & 0xFFFFFFFF for i in range(8)] + [(pt[1] >> (32 * i)) & 0xFFFFFFFF for i in range(8)]  # noqa: E731
    def g2w(pt):
        ws = []
        for v in (pt[0][0], pt[0][1], pt[1][0], pt[1][1]):
            ws += [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
        return ws
    n = 40
    ks = [rnd.randrange(1, b.R) for _ in range(n)]
    signs = [rnd.choice((-1, 1)) for _ in range(n)]
    ko = rnd.randrange(1, b.R)
    total = (2 * ko + sum(s * k for s, k in zip(signs, ks))) % b.R
    arr = (ctypes.c_int * n)(*signs)
    o16, o32 = (ctypes.c_uint32 * 16)(), (ctypes.c_uint32 * 32)()
    pts1 = (ctypes.c_uint32 * (16 * n))(*sum((g1w(b.G1C.mul_pt(b.G1, k)) for k in ks), []))
    lib.emul_g1_lazy_chain((ctypes.c_uint32 * 16)(*g1w(b.G1C.mul_pt(b.G1, ko))), pts1, arr, n, o16)
    assert I(o16).to_bytes(64, "little") == b.ser_g1(b.G1C.mul_pt(b.G1, total))
    lib.emul_g1_xyzz_chain((ctypes.c_uint32 * 16)(*g1w(b.G1C.mul_pt(b.G1, ko))), pts1, arr, n, o16)          # the XYZZ accumulator of the gather kernel
    assert I(o16).to_bytes(64, "little") == b.ser_g1(b.G1C.mul_pt(b.G1, total))
    mx = ctypes.c_uint32()                                                                                    # ... and on nine 29-bit limbs, the form it runs in
    lib.emul_g1_xyzz9_chain((ctypes.c_uint32 * 16)(*g1w(b.G1C.mul_pt(b.G1, ko))), pts1, arr, n, o16, ctypes.byref(mx))
    assert I(o16).to_bytes(64, "little") == b.ser_g1(b.G1C.mul_pt(b.G1, total)) and mx.value < 1 << 29
    pts2 = (ctypes.c_uint32 * (32 * n))(*sum((g2w(b.G2C.mul_pt(b.G2, k)) for k in ks), []))
    lib.emul_g2_lazy_chain((ctypes.c_uint32 * 32)(*g2w(b.G2C.mul_pt(b.G2, ko))), pts2, arr, n, o32)
    assert I(o32).to_bytes(128, "little") == b.ser_g2(b.G2C.mul_pt(b.G2, total))
    lib.emul_g2_xyzz9_chain((ctypes.c_uint32 * 32)(*g2w(b.G2C.mul_pt(b.G2, ko))), pts2, arr, n, o32, ctypes.byref(mx))      # the form k_msm_gather<G2Msm> runs in
    assert I(o32).to_bytes(128, "little") == b.ser_g2(b.G2C.mul_pt(b.G2, total)) and mx.value < 1 << 29


def test_glv_split_of_the_c_element_multiplications(libs):
    """fr_glv_split / jac_mul_u128_signed (bn254_g.h): k = k1 + k2 lambda mod r with |k1|, |k2| < 2^128 for every k < 2^256 the
    floor quotients can meet, and k1 P + k2 phi(P) = k P."""
    lib, _ = libs
    rnd = random.Random(31)
    lam = 0xb3c4d79d41a917585bfc41088d8daaa78b17ea66b99c90dd
    assert (lam * lam + lam + 1) % b.R == 0
    g1w = lambda pt: [(pt[0] >> (32 * i)) & 0xFFFFFFFF for i in range(8)] + [(pt[1] >> (32 * i)) & 0xFFFFFFFF for i in range(8)]  # noqa: E731
    P = b.G1C.mul_pt(b.G1, rnd.randrange(1, b.R))
    pw = (ctypes.c_uint32 * 16)(*g1w(P))
    ks = [0, 1, 2, b.R - 1, b.R - 2, lam, lam + 1, b.R - lam, (1 << 254) - 1, (1 << 128) - 1, 1 << 128, 1 << 127] + [rnd.randrange(b.R) for _ in range(40)] + [rnd.randrange(1 << 64) for _ in range(4)]
    for k in ks:
        halves, out = (ctypes.c_uint32 * 10)(), (ctypes.c_uint32 * 16)()
        lib.emul_g1_glv_mul(pw, W(k), halves, out)
        k1 = sum(int(halves[i]) << (32 * i) for i in range(4)) * (-1 if halves[4] else 1)
        k2 = sum(int(halves[5 + i]) << (32 * i) for i in range(4)) * (-1 if halves[9] else 1)
        assert (k1 + k2 * lam) % b.R == k % b.R
        want = b.G1C.mul_pt(P, 2 * k % b.R)
        assert I(out).to_bytes(64, "little") == b.ser_g1(want), k


def test_key_table_entries_pack_into_eight_words(libs):
    """fq9_pack8 / fq9_unpack8 (bn254_fq9.h): a G1 table entry is 64 bytes, a G2 entry 128.  The eight words are the nine-limb
    integer re-sliced; whatever k_g16_build_table stores (2 x R10 mod-p representatives, < 2.4 p) lies below 2^256 and survives."""
    lib, _ = libs
    rnd = random.Random(29)
    p = b.P
    W8 = ctypes.c_uint32 * 8
    for trial in range(300):
        n = rnd.choice((0, 1, (1 << 256) - 1, rnd.randrange(1 << 256), rnd.randrange(1 << 29), (1 << 232) | rnd.randrange(1 << 232)))
        w = W8(*[(n >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
        limbs, again = (ctypes.c_uint32 * 9)(), W8()
        lib.emul_fq9_pack(w, limbs, again)
        assert all(limbs[i] < 1 << 29 for i in range(8)) and limbs[8] < 1 << 24
        assert sum(int(limbs[i]) << (29 * i) for i in range(9)) == n and list(again) == list(w)
    R9 = 1 << 261
    for trial in range(300):
        a = rnd.choice((0, 1, p - 1, rnd.randrange(p)))
        packed, limbs = W8(), (ctypes.c_uint32 * 9)()
        lib.emul_fq9_entry(W8(*[(a >> (32 * i)) & 0xFFFFFFFF for i in range(8)]), packed, limbs)
        v = sum(int(limbs[i]) << (29 * i) for i in range(9))
        assert v < 3 * p and v < 1 << 256 and v % p == a * R9 % p          # the loop's Montgomery form, inside the packable range
        assert I(packed) == v


def _run(lib, kind, value, the_set, seed, radix_bits=0):
    nv, m = (334, 512) if kind == 0 else (653, 1024)
    z = np.zeros((nv, 8), dtype=np.uint32)
    h = np.zeros((m - 1, 8), dtype=np.uint32)
    rs = np.zeros(16, dtype=np.uint32)
    out = np.zeros(1024, dtype=np.uint8)
    shape = np.zeros(4, dtype=np.uint32)
    sv = np.array(list(the_set) + [0], dtype=np.uint64)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    lib.emul_g16_witness_qap(kind, ctypes.c_uint64(value), P(sv), len(the_set), seed, P(z), P(h), P(rs), P(out), ctypes.c_uint64(1024), P(shape), radix_bits)
    toint = lambda a: [int.from_bytes(r.tobytes(), "little") for r in a]  # noqa: E731
    return toint(z), toint(h), toint(rs.reshape(2, 8)), out.tobytes(), [int(x) for x in shape]


def test_witness_and_quotient_equal_oracle(libs):
    _, lib = libs
    seed = bytes(range(32))
    for value in (42, 0, 2**64 - 1):
        z, h, rs, out, shape = _run(lib, 0, value, [], seed)
        cs = g.equality_circuit(value, value, g.mimc_hash_native(value))
        assert shape == [2, 332, 332, 512]
        assert z == cs.assignment() and h == g.witness_map(cs)[:511]
        assert rs == [g.draw_fr(seed, 0x47313600, 0), g.draw_fr(seed, 0x47313600, 1)]
        assert out[:10] == bytes([2, 2]) + (256).to_bytes(4, "little") + (32).to_bytes(4, "little") and out[266:298] == g.commit_value_snark(value)
    for value, the_set in ((25, [10, 20, 25, 30, 40]), (7, [7]), (9, [7, 9, 9]), (63, list(range(64)))):
        z, h, rs, out, shape = _run(lib, 1, value, the_set, seed)
        sel, sv, ir = g.membership_inputs(value, the_set)
        cs = g.membership_circuit(value, sel, sv, ir, g.mimc_hash_native(value))
        assert shape == [130, 523, 653, 1024] and g.is_satisfied(cs)
        assert z == cs.assignment() and h == g.witness_map(cs)[:1023]
        n = len(the_set)
        assert out[:2] == bytes([2, 4]) and out[10:14] == n.to_bytes(4, "little")
        assert out[14:14 + 8 * n] == b"".join(x.to_bytes(8, "little") for x in the_set)


def test_digit_rows_at_every_key_table_radix(libs):
    """The radix of a key's tables is chosen when the key is loaded (2^14 down to 2^8 by free memory): the witness / QAP steps write
    their digit rows for whatever radix the view carries, and the digits of every radix reassemble to the same h."""
    _, lib = libs
    seed = bytes(range(32))
    want = _run(lib, 0, 42, [], seed)
    for wb in (8, 9, 10, 11, 12, 13, 14, 15, 114):          # 114: radix 2^14 in its uneven form (18 windows), the default when it fits
        got = _run(lib, 0, 42, [], seed, wb)
        assert got == want, wb


def _vk_args(pk):
    g1w = lambda pt: [(pt[0] >> (32 * i)) & 0xFFFFFFFF for i in range(8)] + [(pt[1] >> (32 * i)) & 0xFFFFFFFF for i in range(8)]  # noqa: E731

    def g2w(pt):
        ws = []
        for v in (pt[0][0], pt[0][1], pt[1][0], pt[1][1]):
            ws += [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
        return ws
    A = lambda ws: (ctypes.c_uint32 * len(ws))(*ws)  # noqa: E731
    ic = sum((g1w(p) for p in pk.gamma_abc_g1), [])
    return A(g1w(pk.alpha_g1)), A(g2w(pk.beta_g2)), A(g2w(pk.gamma_g2)), A(g2w(pk.delta_g2)), len(pk.gamma_abc_g1), A(ic)


def test_tower_and_pairing(libs):
    """Fq12 tower arithmetic against the oracle's polynomial-basis Fq12, and bilinearity / non-degeneracy of the ate pairing."""
    lib, _ = libs
    rnd = random.Random(3)
    W12 = lambda a: (ctypes.c_uint32 * 96)(*[(c >> (32 * i)) & 0xFFFFFFFF for c in a for i in range(8)])  # noqa: E731
    I12 = lambda w: [sum(int(w[8 * k + i]) << (32 * i) for i in range(8)) for k in range(12)]  # noqa: E731
    out = (ctypes.c_uint32 * 96)()
    for _ in range(10):
        x, y = [rnd.randrange(b.P) for _ in range(12)], [rnd.randrange(b.P) for _ in range(12)]
        lib.emul_f12_mul(W12(x), W12(y), 0, out)
        assert I12(out) == b.f12_mul(x, y)
        lib.emul_f12_mul(W12(x), W12(y), 1, out)
        assert I12(out) == b.f12_mul(x, x)
    out2 = (ctypes.c_uint32 * 96)()
    for _ in range(3):                                          # inverse; final exponentiation: easy/hard split == one 2790-bit power == oracle
        x = [rnd.randrange(b.P) for _ in range(12)]
        lib.emul_f12_inv(W12(x), out)
        assert I12(out) == b.f12_inv(x)
        lib.emul_f12_final_exp(W12(x), 0, out)
        lib.emul_f12_final_exp(W12(x), 1, out2)
        assert I12(out) == I12(out2)
    assert I12(out) == b.final_exponentiate(x)
    X = 4965661367192848881                                     # addition-chain variant: the same power raised to m = 2x(6x^2 + 3x + 1)
    lib.emul_f12_final_exp_chain(W12(x), out2)
    assert I12(out2) == b.f12_pow(I12(out), 2 * X * (6 * X * X + 3 * X + 1))
    lib.emul_f12_cyclo_sq(W12(x), 0, out)                       # Granger-Scott squaring == general squaring inside the cyclotomic subgroup
    lib.emul_f12_cyclo_sq(W12(x), 1, out2)
    assert I12(out) == I12(out2)
    abc = (ctypes.c_uint32 * 48)(*[(v >> (32 * i)) & 0xFFFFFFFF for v in [rnd.randrange(b.P) for _ in range(6)] for i in range(8)])
    lib.emul_f12_mul_line(W12(x), abc, 0, out)                  # sparse line product == general product
    lib.emul_f12_mul_line(W12(x), abc, 1, out2)
    assert I12(out) == I12(out2)
    for j in (1, 2, 3):
        lib.emul_f12_frob(W12(x), j, out2)
        assert I12(out2) == b.f12_pow(x, b.P ** j)
    g1w = lambda pt: [(pt[0] >> (32 * i)) & 0xFFFFFFFF for i in range(8)] + [(pt[1] >> (32 * i)) & 0xFFFFFFFF for i in range(8)]  # noqa: E731

    def g2w(pt):
        ws = []
        for v in (pt[0][0], pt[0][1], pt[1][0], pt[1][1]):
            ws += [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
        return ws

    def check(pairs):
        g1 = (ctypes.c_uint32 * (16 * len(pairs)))(*sum((g1w(p) for p, q in pairs), []))
        g2 = (ctypes.c_uint32 * (32 * len(pairs)))(*sum((g2w(q) for p, q in pairs), []))
        return lib.emul_pairing_product_is_one(len(pairs), g1, g2)
    a = rnd.randrange(1, b.R)
    Pp, Q = b.G1C.mul_pt(b.G1, rnd.randrange(1, b.R)), b.G2C.mul_pt(b.G2, rnd.randrange(1, b.R))
    aP, aQ, nP = b.G1C.mul_pt(Pp, a), b.G2C.mul_pt(Q, a), b.G1C.neg_pt(Pp)
    assert check([(aP, Q), (nP, aQ)]) == 1                      # e(aP, Q) e(-P, aQ) = 1
    assert check([(aP, Q), (nP, b.G2C.mul_pt(Q, a + 1))]) == 0
    assert check([(Pp, Q)]) == 0                                # non-degenerate


def test_groth16_verifier_verdicts_equal_oracle(libs):
    """g16_verify.h (the GPU verifier's code, on the host) against the oracle's pairing verifier: valid proofs and a
    flipped bit in every region of equality and membership envelopes."""
    _, lib = libs
    SS = bytes(range(32))
    rnd = random.Random(1)
    seed = bytes(range(7, 39))
    r_, s_ = g.draw_fr(seed, 0x47313600, 0), g.draw_fr(seed, 0x47313600, 1)
    pk = g.equality_key(SS)
    va = _vk_args(pk)
    v = 123456789
    cm = g.commit_value_snark(v)
    env = g.envelope(2, g.prove_with_trapdoor(pk, g.equality_circuit(v, v, int.from_bytes(cm, "little")), r_, s_), cm)
    assert lib.emul_g16_verify(0, env, len(env), *va) == 1 and g.verify_equality_with_commitment(env, cm, SS)
    for pos in (1, 5, 12, 80, 150, 210, 265, 270, 297):
        bad = bytearray(env); bad[pos] ^= 1 << rnd.randrange(6 if pos == 265 else 8)     # (265: C's last byte; its top two bits are flags, below)
        assert bool(lib.emul_g16_verify(0, bytes(bad), len(bad), *va)) == g.verify_equality_with_commitment(bytes(bad), bytes(bad[266:]), SS) is False
    # ark-serialize's flag rules: the sign flag of an uncompressed finite point is ignored, both flag bits set is an error
    for last in (73, 201, 265):
        flip = bytearray(env); flip[last] ^= 0x80
        assert lib.emul_g16_verify(0, bytes(flip), len(flip), *va) == 1 and g.verify_equality_with_commitment(bytes(flip), cm, SS)
        both = bytearray(env); both[last] |= 0xC0
        assert lib.emul_g16_verify(0, bytes(both), len(both), *va) == 0 and not g.verify_equality_with_commitment(bytes(both), cm, SS)
    assert lib.emul_g16_verify(0, env[:-1], len(env) - 1, *va) == 0
    # membership
    pkm = g.membership_key(SS)
    vm = _vk_args(pkm)
    the_set = [10, 20, 25, 30, 2**40]
    envm = g.prove_membership(25, the_set, SS, seed)
    assert lib.emul_g16_verify(1, envm, len(envm), *vm) == 1 and g.verify_membership(envm, the_set, SS)
    for pos in (3, 11, 16, 30, 60, 200, 300, len(envm) - 5):
        bad = bytearray(envm); bad[pos] ^= 1 << rnd.randrange(8)
        n = int.from_bytes(bad[10:14], "little")
        emb = [int.from_bytes(bad[14 + 8 * i:22 + 8 * i], "little") for i in range(n)] if 0 < n <= 64 and len(bad) == 10 + 4 + 8 * n + 288 else the_set
        assert bool(lib.emul_g16_verify(1, bytes(bad), len(bad), *vm)) == g.verify_membership(bytes(bad), emb, SS) is False, pos


def test_public_input_point_on_cooperating_lanes_equals_the_one_lane_sum(libs):
    """g16_public_input_lane (what k_g16_public_inputs runs: the table steps of one envelope dealt to 16 lanes, window tables as the device
    builds them) against the one-lane accumulation without tables: equality, membership sets of 1 / 5 / 64 elements with 64-bit extremes,
    other lane counts, and the headers that are refused."""
    _, lib = libs
    SS = bytes(range(32))
    seed = bytes(range(9, 41))
    pk = g.equality_key(SS)
    va = _vk_args(pk)
    v = 2**63 + 12345
    cm = g.commit_value_snark(v)
    r_, s_ = g.draw_fr(seed, 0x47313600, 0), g.draw_fr(seed, 0x47313600, 1)
    env = g.envelope(2, g.prove_with_trapdoor(pk, g.equality_circuit(v, v, int.from_bytes(cm, "little")), r_, s_), cm)
    for lanes in (1, 3, 16, 64):
        assert lib.emul_g16_public_input_lanes(0, env, len(env), va[4], va[5], lanes) == 1
    assert lib.emul_g16_public_input_lanes(0, env[:-1], len(env) - 1, va[4], va[5], 16) == -1
    top = bytearray(env); top[266:298] = (g.R).to_bytes(32, "little")          # commitment = r: not a canonical scalar
    assert lib.emul_g16_public_input_lanes(0, bytes(top), len(top), va[4], va[5], 16) == -1
    pkm = g.membership_key(SS)
    vm = _vk_args(pkm)
    for the_set in ([7], [10, 20, 25, 0, 2**64 - 1], [3 * i + 1 for i in range(63)] + [2**64 - 1]):
        # (the accumulation reads the header, the set and the commitment, not the proof: any 256 bytes do, and no proving is needed here)
        payload = len(the_set).to_bytes(4, "little") + b"".join(x.to_bytes(8, "little") for x in the_set) + bytes(range(256))
        envm = bytes([2, 4]) + len(payload).to_bytes(4, "little") + (32).to_bytes(4, "little") + payload + g.commit_value_snark(the_set[0])
        for lanes in ((16,) if len(the_set) > 5 else (16, 5)):
            assert lib.emul_g16_public_input_lanes(1, envm, len(envm), vm[4], vm[5], lanes) == 1, (len(the_set), lanes)
    bad = bytearray(envm); bad[10] = 65                                          # n past the circuit's 64 slots
    assert lib.emul_g16_public_input_lanes(1, bytes(bad), len(bad), vm[4], vm[5], 16) == -1


def test_one_pairing_check_for_a_batch_accepts_valid_envelopes_and_rejects_a_tampered_one(libs):
    """g16_rlc.h on the host: the weighted batch check (one product of Miller loops on (B_j, rho_j A_j), one virtual envelope for gamma, delta
    and beta) accepts envelopes the oracle's verifier accepts, rejects the batch when one commitment / one proof element is swapped for
    another valid one, and hands envelopes it does not take (refused header, point at infinity) back to the per-envelope path."""
    _, lib = libs
    SS = bytes(range(32))
    rnd = random.Random(5)
    pk = g.equality_key(SS)
    va = _vk_args(pk)
    envs = []
    for k in range(3):
        seed = bytes([k + 1]) * 32
        v = rnd.randrange(2**64)
        cm = g.commit_value_snark(v)
        envs.append(g.envelope(2, g.prove_with_trapdoor(pk, g.equality_circuit(v, v, int.from_bytes(cm, "little")), g.draw_fr(seed, 0x47313600, 0), g.draw_fr(seed, 0x47313600, 1)), cm))
        assert g.verify_equality_with_commitment(envs[-1], cm, SS)

    def check(kind, batch, args):
        n = len(batch); stride = max(len(e) for e in batch)
        buf = b"".join(e + bytes(stride - len(e)) for e in batch)
        lens = (ctypes.c_uint32 * n)(*[len(e) for e in batch])
        rho = (ctypes.c_uint32 * (4 * n))(*[rnd.randrange(1, 2**32) for _ in range(4 * n)])
        return lib.emul_g16_rlc(kind, n, buf, stride, lens, rho, *args)

    assert check(0, envs, va) == 1
    assert check(0, envs[:1], va) == 1
    swapped = [envs[0][:266] + envs[1][266:], envs[1], envs[2]]                      # a valid proof under another envelope's commitment
    assert check(0, swapped, va) == 0
    crossed = [envs[0][:10] + envs[1][10:74] + envs[0][74:], envs[1], envs[2]]       # A of another (valid) proof
    assert check(0, crossed, va) == 0
    inf = bytearray(envs[2]); inf[10:74] = bytes(63) + bytes([0x40])                  # A at infinity: a valid encoding the fast path does not take
    assert check(0, [envs[0], bytes(inf)], va) == 2
    assert check(0, [envs[0], envs[1][:-1]], va) == 2
    pkm = g.membership_key(SS)
    vm = _vk_args(pkm)
    m1 = g.prove_membership(25, [10, 20, 25, 30, 2**40], SS, bytes(range(3, 35)))
    m2 = g.prove_membership(2**64 - 1, [2**64 - 1, 0, 7], SS, bytes(range(4, 36)))
    assert check(1, [m1, m2], vm) == 1
    bad = bytearray(m2); bad[14 + 8] ^= 1                                              # a set element of the second envelope
    assert check(1, [m1, bytes(bad)], vm) == 0


def test_fq2_machine_verdicts_equal_the_lane_per_chain_verifier(libs):
    """fq2vm.h + the generated tables (tools/gen_fq2vm.py), executed on the host round by round (four waves per chain, two
    half-waves each), against g16_verify.h and the oracle on the same envelopes: valid, a flipped bit in every region, ark's flag rules."""
    _, lib = libs
    SS = bytes(range(32))
    rnd = random.Random(3)
    seed = bytes(range(9, 41))
    r_, s_ = g.draw_fr(seed, 0x47313600, 0), g.draw_fr(seed, 0x47313600, 1)
    pk = g.equality_key(SS)
    va = _vk_args(pk)
    v = 987654321
    cm = g.commit_value_snark(v)
    env = g.envelope(2, g.prove_with_trapdoor(pk, g.equality_circuit(v, v, int.from_bytes(cm, "little")), r_, s_), cm)
    cases = [env]
    for pos in (1, 12, 40, 80, 100, 150, 210, 240, 270, 297):
        bad = bytearray(env); bad[pos] ^= 1 << rnd.randrange(6); cases.append(bytes(bad))
    for last in (73, 201, 265):
        flip = bytearray(env); flip[last] ^= 0x80; cases.append(bytes(flip))
    inf = bytearray(env); inf[10:74] = bytes(63) + b"\x40"; cases.append(bytes(inf))          # A = the point at infinity: left to the other path
    for waves in (4,):
        seen = set()
        for e in cases:
            want = lib.emul_g16_verify(0, e, len(e), *va)
            got = lib.emul_g16_verify_vm(waves, 0, e, len(e), *va)
            assert got == 2 or got == want, (waves, e.hex())
            seen.add(got)
        assert seen == {0, 1, 2}
    assert g.verify_equality_with_commitment(env, cm, SS)
    pkm = g.membership_key(SS)
    vm = _vk_args(pkm)
    envm = g.prove_membership(25, [10, 20, 25, 30, 2**40], SS, seed)
    assert lib.emul_g16_verify_vm(4, 1, envm, len(envm), *vm) == 1
    bad = bytearray(envm); bad[200] ^= 4
    assert lib.emul_g16_verify_vm(4, 1, bytes(bad), len(bad), *vm) == lib.emul_g16_verify(1, bytes(bad), len(bad), *vm) == 0


def test_fq2_machine_generator_self_check():
    """tools/gen_fq2vm.py --check: the traced formulas against the oracle's pairing (bilinearity, non-degeneracy, a twist point outside
    the subgroup), and every schedule / register allocation against the traced program; the committed header is what the tool emits."""
    import subprocess, sys, os, tempfile
    tool = os.path.join(ROOT, "tools", "gen_fq2vm.py")
    out = subprocess.run([sys.executable, tool, "--check"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "schedules ok" in out.stdout, out.stdout + out.stderr
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_fq2vm
    with tempfile.TemporaryDirectory() as d:
        gen_fq2vm.emit(os.path.join(d, "h"))
        assert open(os.path.join(d, "h")).read() == open(os.path.join(ROOT, "libzkp_amd", "csrc", "fq2vm_programs.h")).read()
