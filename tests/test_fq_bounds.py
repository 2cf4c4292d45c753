"""Interval-arithmetic proof for the unsaturated BN254 base field (libzkp_amd/csrc/bn254_fq.h) and the lazily reduced
mixed additions of the Groth16 MSM inner loops (bn254_g.h: g1_madd_lazy, g2_madd_lazy): no 32-bit limb operation and
no 64-bit column sum overflows, every borrowed-multiple subtraction has a covered subtrahend, and every result handed
to the next iteration is "safe" again (no GPU).

A tracked value is (limbs, val): inclusive upper bounds of the ten limbs and of the integer value, in units of p/1000."""
P = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47
R_OVER_P = (1 << 260) * 1000 // P            # 84.6 (x1000)
PL = [(P >> (26 * i)) & 0x3FFFFFF if i < 9 else P >> 234 for i in range(10)]
M26 = (1 << 26) - 1


def header_constants():
    import os
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "libzkp_amd", "csrc", "bn254_fq.h")).read()
    out = {}
    for name in ("fq_pl", "fq_k4", "fq_k8", "fq_k16", "fq_one_l", "fq_r2_l"):
        m = re.search(name + r"\(int i\) \{ constexpr uint32_t m\[10\] = \{([^}]*)\}", src)
        out[name] = [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]
    out["n0"] = int(re.search(r"#define ZKP_FQ_N0 (0x[0-9a-f]+)u", src).group(1), 16)
    out["recip"] = int(re.search(r"#define ZKP_FQ_RECIP (\d+)u", src).group(1))
    return out


C = header_constants()


def test_constants():
    val = lambda l: sum(x << (26 * i) for i, x in enumerate(l))  # noqa: E731
    assert C["fq_pl"] == PL
    for n, name in ((4, "fq_k4"), (8, "fq_k8"), (16, "fq_k16")):
        assert val(C[name]) == n * P
        assert all(x >= n << 24 for x in C[name][:9]) and all(x < 1 << 32 for x in C[name])
    assert val(C["fq_one_l"]) == (1 << 260) % P and val(C["fq_r2_l"]) == (1 << 520) % P
    assert (C["n0"] * P + 1) % (1 << 26) == 0
    assert C["recip"] == (1 << 266) // P


# ---- tracked values
class V:
    def __init__(self, limbs, val):
        self.l, self.val = list(limbs), val
        assert all(x < 1 << 32 for x in self.l), "limb overflows 32 bits"


def carried(val_milli_p):
    top = (val_milli_p * P // 1000) >> 234
    return V([M26] * 9 + [top], val_milli_p)


SAFE = carried(3000)          # what fq_reduce_weak returns
CANON = carried(2000)         # table entries: fq_mul outputs < 2p


def add_l(a, b):
    return V([x + y for x, y in zip(a.l, b.l)], a.val + b.val)


def dbl_l(a):
    return add_l(a, a)


def sub_k(n, a, b):
    k = C["fq_k%d" % n]
    assert all(y <= kk for y, kk in zip(b.l, k)), ("subtrahend limb exceeds borrowed %dp" % n, b.l, k)
    return V([x + kk for x, kk in zip(a.l, k)], a.val + 1000 * n)


def mul(a, b):
    for col in range(19):
        acc = sum(a.l[j] * b.l[col - j] for j in range(10) if 0 <= col - j < 10)
        acc += sum(M26 * PL[col - j] for j in range(10) if 0 <= col - j < 10)
        assert acc + (1 << 38) < 1 << 64, ("column", col, acc.bit_length())
    out = a.val * b.val // R_OVER_P + 1000 + 1
    assert out < 4000, "product leaves the safe range"
    return carried(out)


def sq(a):
    """fq_sq: pairs are formed once with a doubled limb; the column sums are term for term those of mul(a, a)"""
    assert all(2 * x < 1 << 32 for x in a.l), "doubled limb overflows 32 bits"
    for col in range(19):
        acc = sum(2 * a.l[j] * a.l[col - j] for j in range(10) if 0 <= col - j < 10 and 2 * j < col)
        acc += a.l[col // 2] ** 2 if col % 2 == 0 else 0
        full = sum(a.l[j] * a.l[col - j] for j in range(10) if 0 <= col - j < 10)
        assert acc == full
    return mul(a, a)


def reduce_weak(a):
    assert a.val * P // 1000 < 1 << 260
    assert all(x + (1 << 6) < 1 << 32 for x in a.l)
    return carried(3000)


def carry(a):
    assert a.val * P // 1000 < 1 << 260 and all(x + (1 << 6) < 1 << 32 for x in a.l)
    return carried(a.val)


def g1_madd_lazy(X1, Y1, Z1, x2, y2):
    Z1Z1 = sq(Z1)
    U2, S2 = mul(x2, Z1Z1), mul(mul(y2, Z1), Z1Z1)
    H, sv = sub_k(4, U2, X1), sub_k(4, S2, Y1)
    HH = sq(H)
    I = dbl_l(dbl_l(HH))
    J, Vv = mul(H, I), mul(X1, I)
    ss = sq(sv)
    X3 = reduce_weak(sub_k(8, sub_k(4, dbl_l(dbl_l(ss)), J), dbl_l(Vv)))
    t = sub_k(4, mul(sv, sub_k(4, Vv, X3)), mul(Y1, J))
    Y3 = reduce_weak(dbl_l(t))
    Z3 = reduce_weak(sub_k(4, sub_k(4, sq(add_l(Z1, H)), Z1Z1), HH))
    return X3, Y3, Z3


def test_g1_madd_lazy_is_closed_over_safe_inputs():
    zero = V([0] * 10, 0)
    neg_y = sub_k(4, zero, CANON)                       # 4p - y for a negative digit
    for y2 in (CANON, neg_y):
        out = g1_madd_lazy(SAFE, SAFE, SAFE, CANON, y2)
        assert all(o.val <= 3000 and max(o.l[:9]) <= M26 for o in out)


def mul_add2(a, b, c, d):
    """fq_mul_add2: both products in the same columns, one reduction"""
    for col in range(19):
        acc = sum(a.l[j] * b.l[col - j] + c.l[j] * d.l[col - j] for j in range(10) if 0 <= col - j < 10)
        acc += sum(M26 * PL[col - j] for j in range(10) if 0 <= col - j < 10)
        assert acc + (1 << 38) < 1 << 64, ("column", col, acc.bit_length())
    out = (a.val * b.val + c.val * d.val) // R_OVER_P + 1000 + 1
    assert out < 4000, "product leaves the safe range"
    return carried(out)


def g1_mmadd_lazy(X1, Y1, ZZ1, ZZZ1, x2, y2):
    U2, S2 = mul(x2, ZZ1), mul(y2, ZZZ1)
    Pv, Rv = sub_k(4, U2, X1), sub_k(4, S2, Y1)
    PP = sq(Pv)
    PPP, Q = mul(Pv, PP), mul(X1, PP)
    RR = sq(Rv)
    X3 = reduce_weak(sub_k(8, sub_k(4, RR, PPP), dbl_l(Q)))
    zero = V([0] * 10, 0)
    Y3 = mul_add2(Rv, sub_k(4, Q, X3), sub_k(4, zero, Y1), PPP)
    return X3, Y3, mul(ZZ1, PP), mul(ZZZ1, PPP)


def test_g1_mmadd_lazy_is_closed_over_safe_inputs():
    """the XYZZ accumulator of k_msm_gather<G1Msm>: every coordinate handed to the next iteration is safe again, and so are the
    coordinates of the Jacobian point it is converted to at the end of a chunk"""
    zero = V([0] * 10, 0)
    neg_y = sub_k(4, zero, CANON)
    for y2 in (CANON, neg_y):
        out = g1_mmadd_lazy(SAFE, SAFE, SAFE, SAFE, CANON, y2)
        assert all(o.val <= 3000 and max(o.l[:9]) <= M26 for o in out)
    reduce_weak(mul(SAFE, SAFE)); reduce_weak(SAFE)                      # jac_from_xyzz
    sq(SAFE); mul(sq(SAFE), SAFE)                                        # xyzz_from_jac


def kara(a, b):
    return mul(a[0], b[0]), mul(a[1], b[1]), mul(add_l(a[0], a[1]), add_l(b[0], b[1]))


def join(k):
    return sub_k(4, k[0], k[1]), sub_k(8, k[2], add_l(k[0], k[1]))


def sq_l(a):
    return mul(add_l(a[0], a[1]), sub_k(4, a[0], a[1])), mul(a[0], a[1])


def g2_madd_lazy(X1, Y1, Z1, x2, y2):
    zz = sq_l(Z1)
    Z1Z1 = (zz[0], dbl_l(zz[1]))
    U2 = join(kara(x2, Z1Z1))
    S2 = join(kara(join(kara(y2, Z1)), Z1Z1))
    H = tuple(reduce_weak(sub_k(4, U2[i], X1[i])) for i in range(2))
    sv = tuple(reduce_weak(sub_k(4, S2[i], Y1[i])) for i in range(2))
    hh = sq_l(H)
    I = (dbl_l(dbl_l(hh[0])), dbl_l(dbl_l(dbl_l(hh[1]))))
    J, Vk = kara(H, I), kara(X1, I)
    ss = sq_l(sv)
    X3 = (reduce_weak(sub_k(16, add_l(dbl_l(dbl_l(ss[0])), add_l(J[1], dbl_l(Vk[1]))), add_l(J[0], dbl_l(Vk[0])))),
          reduce_weak(sub_k(16, add_l(dbl_l(dbl_l(dbl_l(ss[1]))), add_l(add_l(J[0], J[1]), dbl_l(add_l(Vk[0], Vk[1])))), add_l(J[2], dbl_l(Vk[2])))))
    Wv = (sub_k(8, Vk[0], add_l(Vk[1], X3[0])), sub_k(16, Vk[2], add_l(add_l(Vk[0], Vk[1]), X3[1])))
    P1, P2 = kara(sv, Wv), kara(Y1, join(J))
    Y3 = (reduce_weak(dbl_l(sub_k(8, add_l(P1[0], P2[1]), add_l(P1[1], P2[0])))),
          reduce_weak(dbl_l(sub_k(16, add_l(P1[2], add_l(P2[0], P2[1])), add_l(add_l(P1[0], P1[1]), P2[2])))))
    zh = (add_l(Z1[0], H[0]), add_l(Z1[1], H[1]))
    zq0, zqm = mul(add_l(zh[0], zh[1]), sub_k(8, zh[0], zh[1])), mul(zh[0], zh[1])
    Z3 = (reduce_weak(sub_k(8, zq0, add_l(zz[0], hh[0]))), reduce_weak(dbl_l(sub_k(8, zqm, add_l(zz[1], hh[1])))))
    return X3 + Y3 + Z3


def test_g2_madd_lazy_is_closed_over_safe_inputs():
    zero = V([0] * 10, 0)
    s2, c2 = (SAFE, SAFE), (CANON, CANON)
    neg = (sub_k(4, zero, CANON), sub_k(4, zero, CANON))
    for y2 in (c2, neg):
        out = g2_madd_lazy(s2, s2, s2, c2, y2)
        assert all(o.val <= 3000 and max(o.l[:9]) <= M26 for o in out)


def test_reduce_weak_quotient():
    """q = floor(top * RECIP / 2^32) never exceeds floor(x / p) and leaves x - q p < 3p (exhaustive over the top limb edges)."""
    import random
    rnd = random.Random(5)
    recip = C["recip"]
    for _ in range(20000):
        x = rnd.randrange(1 << 260) if rnd.random() < 0.7 else rnd.randrange(64) * P + rnd.choice((0, 1, P - 1, P // 2))
        x = min(x, (1 << 260) - 1)
        q = ((x >> 234) * recip) >> 32
        assert 0 <= x - q * P < 3 * P


def test_safe_ops_used_by_generic_point_formulas():
    """f_add / f_sub / f_dbl (reduce_weak after one limb-wise op) and the Karatsuba Fq2 product on safe inputs."""
    a = SAFE
    reduce_weak(add_l(a, a)); reduce_weak(sub_k(8, a, a)); reduce_weak(dbl_l(a))
    t0, t1 = mul(a, a), mul(a, a)
    t2 = mul(add_l(a, a), add_l(a, a))
    reduce_weak(sub_k(4, t0, t1)); reduce_weak(sub_k(8, t2, add_l(t0, t1)))
    reduce_weak(mul(add_l(a, a), sub_k(4, a, a))); reduce_weak(dbl_l(mul(a, a)))


# ---- the nine-limb form of the G1 MSM loop (bn254_fq9.h): operands are carried (limbs < 2^29), values tracked in units of p
def header_constants9():
    import os
    import re
    src = open(os.path.join(os.path.dirname(__file__), "..", "libzkp_amd", "csrc", "bn254_fq9.h")).read()
    out = {}
    for name in ("fq9_pl", "fq9_k2", "fq9_k4", "fq9_k8", "fq9_r10"):
        m = re.search(name + r"\(int i\) \{ constexpr uint32_t m\[9\] = \{([^}]*)\}", src)
        out[name] = [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]
    out["n0"] = int(re.search(r"#define ZKP_FQ9_N0 (0x[0-9a-f]+)u", src).group(1), 16)
    return out


def test_nine_limb_constants_and_column_sums():
    import os
    c = header_constants9()
    val = lambda l: sum(x << (29 * i) for i, x in enumerate(l))  # noqa: E731
    assert val(c["fq9_pl"]) == P and val(c["fq9_k2"]) == 2 * P and val(c["fq9_k4"]) == 4 * P and val(c["fq9_k8"]) == 8 * P
    assert all(x < 1 << 29 for name in ("fq9_pl", "fq9_k2", "fq9_k4", "fq9_k8", "fq9_r10") for x in c[name])
    assert val(c["fq9_r10"]) == (1 << 260) % P and (c["n0"] * P + 1) % (1 << 29) == 0
    m29 = (1 << 29) - 1
    # widest column of a product: 9 operand products + 9 reduction products + the carry of the column before
    col = 9 * m29 * m29 + 9 * m29 * max(c["fq9_pl"])
    assert col + (col >> 29) < 1 << 64
    # fq9_sq: doubled limbs (< 2^30) in the 4 mixed products of the widest column, one square, 8 reduction products
    assert 4 * (2 * m29) * m29 + m29 * m29 + 9 * m29 * max(c["fq9_pl"]) + (col >> 29) < 1 << 64
    # fq9_mul_add2: 18 operand products + 9 reduction products
    col2 = 18 * m29 * m29 + 9 * m29 * max(c["fq9_pl"])
    assert col2 + (col2 >> 29) < 1 << 64
    # round 4: the fused double product of Y3 takes two LOOSE operands, limb-wise differences against fat forms of 8p / 4p (fq9_sub_loose,
    # fq9_neg_loose): the fat forms are the same integers, no limb difference goes negative, and the widest column still fits 64 bits
    src = open(os.path.join(os.path.dirname(__file__), "..", "libzkp_amd", "csrc", "bn254_fq9.h")).read()
    assert "i == 0 ? fq9_kp<K>(0) + (1u << 29) : i < 8 ? fq9_kp<K>(i) + (1u << 29) - 1u : fq9_kp<K>(8) - 1u" in src
    for name, k in (("fq9_k8", 8), ("fq9_k4", 4)):
        fat = [c[name][0] + (1 << 29)] + [c[name][i] + (1 << 29) - 1 for i in range(1, 8)] + [c[name][8] - 1]
        assert val(fat) == k * P
        assert all(f >= m29 for f in fat[:8]) and all(f < 1 << 30 for f in fat) and fat[8] > 0       # a_i + fat_i - b_i >= 0 for carried a, b (limbs <= 2^29 - 1)
    loose8, loose4 = m29 + (1 << 30) - 1, (1 << 30) - 1                                               # limbs of Q - X3 + 8p and of 4p - Y1
    col_l = 9 * m29 * loose8 + 9 * loose4 * m29 + 9 * m29 * max(c["fq9_pl"])
    assert col_l + (col_l >> 29) < 1 << 64
    # the top limb of a loose difference stays positive: the VALUE Q - X3 + 8p is at least 8p - X3 > 2.8 p (X3 < 5.2 p, test below), and the
    # lower limbs hold at most 2 units more than their carried share each
    assert (28 * P // 10) >> (29 * 8) > 4
    # fq9_sub_k / fq9_sub2_k4: limb expressions fit int32 with the running carry
    assert m29 + max(c["fq9_k8"]) + 1 < 1 << 31 and -(m29 + 2 * m29 + 4) > -(1 << 31)


def test_g1_mmadd9_is_closed_over_its_value_bounds():
    """g1_mmadd9 (bn254_g.h) in units of p: accumulator X < 8, Y < 4, ZZ, ZZZ < 2 and entries < 4 give the same bounds back, every
    subtrahend is covered by the multiple of p added, and nothing comes near 2^261 = 169.28 p (so top limbs stay below 2^29)"""
    from fractions import Fraction as F
    rp = F(1 << 261, P)
    mul = lambda a, b: a * b / rp + 1  # noqa: E731
    X, Y, ZZ, ZZZ, qx, qy = F(8), F(4), F(3), F(3), F(4), F(4)                 # ZZ, ZZZ < 3 covers the converted accumulator a chunk starts from
    U2, S2 = mul(qx, ZZ), mul(qy, ZZZ)
    assert X <= 8 and Y <= 4                                  # fq9_sub_k<8>(U2, X), fq9_sub_k<4>(S2, Y)
    Pv, Rv = U2 + 8, S2 + 4
    PP = mul(Pv, Pv); PPP, Q, RR = mul(Pv, PP), mul(X, PP), mul(Rv, Rv)
    assert PPP + 2 * Q < 4                                    # fq9_sub2_k4
    X3 = RR + 4
    assert X3 < 8                                             # fq9_sub_k<8>(Q, X3), and the next iteration's X
    Y3 = (Rv * (Q + 8) + 4 * PPP) / rp + 1
    ZZ3, ZZZ3 = mul(ZZ, PP), mul(ZZZ, PPP)
    assert Y3 < 4 and ZZ3 < 2 and ZZZ3 < 2
    assert max(Pv, Rv, X3, Q + 8) < rp / 8
    # conversions: fq9_from_fq of a reduce_weak value (< 3p) starts inside the bounds; entries negated as 4p - y with y < 3p
    assert 3 < 4 and 3 <= X and 3 <= Y and 3 <= ZZ
    # key-table entries (k_g16_build_table): an affine coordinate is an fq_mul output of safe operands (< 4p each: < 16 p / 84.6 + p
    # = 1.19 p in ten-limb form), doubled by fq9_from_fq: < 2.4 p < 2^255 -- so the nine-limb integer fits the eight packed words of
    # fq9_pack8 (its top limb stays below 2^24), and after fq9_unpack8 it is inside the entry bound above
    entry = 2 * (F(16 * P, 1 << 260) + 1)
    assert entry < F(24, 10) and entry * P < 1 << 255 and entry < qx


def test_g2_mmadd9_is_closed_over_its_value_bounds():
    """g2_mmadd9 (bn254_g.h), per Fq component in units of p: X < 10.4, Y < 4, ZZ, ZZZ < 3 and entries < 3 give X, Y within the same
    bounds and ZZ, ZZZ < 2; every subtrahend / negated operand is covered by the multiple of p used; nothing comes near 2^261."""
    from fractions import Fraction as F
    rp = F(1 << 261, P)
    m2 = lambda a, b, c, d: (a * b + c * d) / rp + 1  # noqa: E731
    X, Y, ZZ, ZZZ, q = F(104, 10), F(4), F(3), F(3), F(3)
    assert ZZ < 4 and ZZZ < 4                                   # nZZ1, nZZZ1 = 4p - .
    U2 = max(m2(q, ZZ, q, 4), m2(q, ZZ, q, ZZ)); S2 = max(m2(q, ZZZ, q, 4), m2(q, ZZZ, q, ZZZ))
    assert X < 16 and S2 + Y < 8                                # P = U2 - X + 16p ; R = +-S2 - Y + 8p
    Pv, Rv = U2 + 16, S2 + 8
    assert Pv < 32                                              # nP1 = 32p - P1
    PP = max(m2(Pv, Pv, Pv, 32), 2 * Pv * Pv / rp + 1)
    assert PP < 8                                               # nPP1 = 8p - PP1
    PPP = max(m2(Pv, PP, Pv, 8), m2(Pv, PP, Pv, PP)); Q = max(m2(X, PP, X, 8), m2(X, PP, X, PP))
    assert Rv < 16                                              # nR1 = 16p - R1
    RR = max(m2(Rv, Rv, Rv, 16), 2 * Rv * Rv / rp + 1)
    assert PPP + 2 * Q < 8                                      # X3 = RR - PPP - 2Q + 8p
    X3 = RR + 8
    assert X3 <= X and X3 < 16                                  # W = Q - X3 + 16p
    W = Q + 16
    assert Y <= 4                                               # nY0, nY1 = 4p - .
    Y3 = max((Rv * W + 16 * W + 4 * PPP + Y * PPP) / rp + 1, (Rv * W + Rv * W + 4 * PPP + 4 * PPP) / rp + 1)
    ZZ3 = max(m2(PP, ZZ, PP, 4), m2(PP, ZZ, PP, ZZ)); ZZZ3 = max(m2(PPP, ZZZ, PPP, 4), m2(PPP, ZZZ, PPP, ZZZ))
    assert Y3 < 4 and ZZ3 < 2 and ZZZ3 < 2
    assert max(Pv, W, 32) < rp / 4
    # fq9_mul_add4 columns: 36 operand products + 9 reduction products of carried limbs
    m29 = (1 << 29) - 1
    col = 36 * m29 * m29 + 9 * m29 * m29
    assert col + (col >> 29) < 1 << 64
    # fq9_mul with one limb-wise doubled operand (fq2_9_sq): 9 products of < 2^59 + 9 reduction products
    col = 9 * (2 * m29) * m29 + 9 * m29 * m29
    assert col + (col >> 29) < 1 << 64


def test_fr9_constants_and_ntt_growth():
    """bn254_fr9.h: constants, column sums, and the value bounds of the QAP step's transforms (units of r)"""
    import os
    import re
    from fractions import Fraction as F
    R_ = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
    src = open(os.path.join(os.path.dirname(__file__), "..", "libzkp_amd", "csrc", "bn254_fr9.h")).read()
    c = {}
    for name in ("fr9_pl", "fr9_k2", "fr9_k4", "fr9_k32", "fr9_one", "fr9_r256"):
        m = re.search(name + r"\(int i\) \{ constexpr uint32_t m\[9\] = \{([^}]*)\}", src)
        c[name] = [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]
    val = lambda l: sum(x << (29 * i) for i, x in enumerate(l))  # noqa: E731
    assert val(c["fr9_pl"]) == R_ and val(c["fr9_k2"]) == 2 * R_ and val(c["fr9_k4"]) == 4 * R_ and val(c["fr9_k32"]) == 32 * R_
    assert val(c["fr9_one"]) == (1 << 261) % R_ and val(c["fr9_r256"]) == (1 << 256) % R_
    assert all(x < 1 << 29 for v in c.values() for x in v)
    n0 = int(re.search(r"#define ZKP_FR9_N0 (0x[0-9a-f]+)u", src).group(1), 16)
    assert (n0 * R_ + 1) % (1 << 29) == 0
    assert int(re.search(r"#define ZKP_FR9_RECIP (\d+)u", src).group(1)) == (1 << 264) // R_
    m29 = (1 << 29) - 1
    col = 9 * m29 * m29 + 9 * m29 * max(c["fr9_pl"])
    assert col + (col >> 29) < 1 << 64
    rp = F(1 << 261, R_)
    mul = lambda a, b: a * b / rp + 1  # noqa: E731
    tab = mul(64, 1)                                  # fr9_from_fr: 32 * (value < 2r) times R9 mod r (< r)
    assert tab < F(14, 10)
    # The witness map (g16_steps.h: g16_qap_proof): load -> DIF inverse -> scale -> DIT forward -> pointwise -> DIF inverse -> coset_inv.
    # decimation in frequency (first and last transform): reduce_weak(u + v) < 2.3, (u - v + 4r) w; inputs: loaded elements (< 1.4) / the
    # pointwise result (< 1.3)
    e = F(23, 10)
    assert tab < e and e < 4 and mul(e + 4, tab) < e and 2 * e < rp
    scaled = mul(e, tab)                              # the scale step: (element < 2.3 r) * coset table entry
    assert scaled < F(11, 10)
    # decimation in time, ten stages from a scaled element: u + v w and u - v w + 2r
    b = scaled
    for _ in range(10):
        v = mul(b, tab)
        assert v < 2                                  # fr9_sub_k<2>
        b = b + 2
    assert b < 22
    # pointwise: (a b - c + 32 r) zinv with a, b, c outputs of the ten DIT stages
    ab = mul(b, b)
    assert ab < 4 and b < 32 and mul(ab + 32, tab) < F(13, 10) and mul(ab + 32, tab) < e
    # back to eight words: fr9_to_fr of (element < 2.3 r) * coset_inv stays below 2r
    assert mul(mul(e, tab), 1) < 2
    # fr9_reduce_weak: q = floor(top * floor(2^264 / r) / 2^32) never exceeds floor(a / r) and misses it by less than 1.3
    recip = (1 << 264) // R_
    for a in (R_ - 1, 2 * R_, 5 * R_ + 12345, (1 << 261) - 1, 44 * R_ - 1):
        q = ((a >> 232) * recip) >> 32
        assert q <= a // R_ and a - q * R_ < 23 * R_ // 10
