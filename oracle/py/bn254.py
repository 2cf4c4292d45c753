"""ORACLE (test infrastructure only -- never imported by the product path).

BN254 (alt_bn128) arithmetic in Python bigints: Fq, Fr, Fq2, Fq12, G1, G2 and the optimal-ate pairing.
The reference reaches this through the third-party crates ark-bn254 / ark-ec / ark-ff ^0.5
(Cargo.toml:16-21; not vendored, unpinned, unbuildable here).  Call sites: /root/reference/src/backend/snark.rs:4-12.
Published parameters restated: EIP-196/197 (curve, generators), Barreto-Naehrig optimal ate pairing.
Pins (tests/test_oracle_bn254.py): generators on curve and of order r, bilinearity e(aP, bQ) = e(P, Q)^(ab),
non-degeneracy, the EIP-197 style product check e(P, Q) * e(-P, Q) = 1.
"""

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583   # base field
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617   # scalar field (group order)
ATE_LOOP_COUNT = 29793968203157093288
LOG_ATE_LOOP_COUNT = 63


def inv(a, m):
    return pow(a, -1, m)


# ---------------------------------------------------------------- Fq2 = Fq[u]/(u^2 + 1), elements as (c0, c1)
def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_neg(a):
    return (-a[0] % P, -a[1] % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_sq(a):
    return f2_mul(a, a)


def f2_scalar(a, k):
    return (a[0] * k % P, a[1] * k % P)


def f2_inv(a):
    d = inv((a[0] * a[0] + a[1] * a[1]) % P, P)
    return (a[0] * d % P, -a[1] * d % P)


F2_ZERO, F2_ONE = (0, 0), (1, 0)
B1 = 3
B2 = f2_mul((3, 0), f2_inv((9, 1)))     # twist curve coefficient 3/(9+u)


# ---------------------------------------------------------------- generic short-Weierstrass (a = 0) affine arithmetic
class Curve:
    def __init__(self, add, sub, mul, sq, inv_, neg, zero, one, b, scalar):
        self.add, self.sub, self.mul, self.sq, self.inv, self.neg = add, sub, mul, sq, inv_, neg
        self.zero, self.one, self.b, self.scalar = zero, one, b, scalar

    def is_on_curve(self, pt):
        if pt is None:
            return True
        x, y = pt
        return self.sq(y) == self.add(self.mul(self.sq(x), x), self.b)

    def double(self, pt):
        if pt is None:
            return None
        x, y = pt
        if y == self.zero:
            return None
        m = self.mul(self.scalar(self.sq(x), 3), self.inv(self.scalar(y, 2)))
        nx = self.sub(self.sq(m), self.scalar(x, 2))
        return (nx, self.sub(self.mul(m, self.sub(x, nx)), y))

    def add_pts(self, p1, p2):
        if p1 is None:
            return p2
        if p2 is None:
            return p1
        if p1[0] == p2[0]:
            return self.double(p1) if p1[1] == p2[1] else None
        m = self.mul(self.sub(p2[1], p1[1]), self.inv(self.sub(p2[0], p1[0])))
        nx = self.sub(self.sub(self.sq(m), p1[0]), p2[0])
        return (nx, self.sub(self.mul(m, self.sub(p1[0], nx)), p1[1]))

    def neg_pt(self, pt):
        return None if pt is None else (pt[0], self.neg(pt[1]))

    def mul_pt(self, pt, k, reduce=True):
        if reduce:
            k %= R
        acc, base = None, pt
        while k:
            if k & 1:
                acc = self.add_pts(acc, base)
            base = self.double(base)
            k >>= 1
        return acc

    def msm(self, scalars, pts):
        acc = None
        for k, pt in zip(scalars, pts):
            if k % R and pt is not None:
                acc = self.add_pts(acc, self.mul_pt(pt, k))
        return acc


G1C = Curve(lambda a, b: (a + b) % P, lambda a, b: (a - b) % P, lambda a, b: a * b % P, lambda a: a * a % P,
            lambda a: inv(a, P), lambda a: -a % P, 0, 1, B1, lambda a, k: a * k % P)
G2C = Curve(f2_add, f2_sub, f2_mul, f2_sq, f2_inv, f2_neg, F2_ZERO, F2_ONE, B2, f2_scalar)

G1 = (1, 2)
G2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
       11559732032986387107991004021392285783925812861821192530917403151452391805634),
      (8495653923123431417604973247489272438418190587263600148770280649306958101930,
       4082367875863433681332203403145435568316851327593401208105741076214120093531))
assert G1C.is_on_curve(G1) and G2C.is_on_curve(G2)


# ---------------------------------------------------------------- Fq12 = Fq[w]/(w^12 - 18 w^6 + 82), coefficient lists
F12_MOD = [82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0]


def f12(coeffs):
    return [c % P for c in coeffs] + [0] * (12 - len(coeffs))


F12_ONE = f12([1])


def f12_add(a, b):
    return [(x + y) % P for x, y in zip(a, b)]


def f12_sub(a, b):
    return [(x - y) % P for x, y in zip(a, b)]


def f12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for i in range(22, 11, -1):          # reduce by w^12 = 18 w^6 - 82
        c = t[i]
        if c:
            t[i - 6] += 18 * c
            t[i - 12] -= 82 * c
    return [x % P for x in t[:12]]


def f12_scalar(a, k):
    return [x * k % P for x in a]


def f12_pow(a, e):
    acc, base = F12_ONE, a
    while e:
        if e & 1:
            acc = f12_mul(acc, base)
        base = f12_mul(base, base)
        e >>= 1
    return acc


def _poly_deg(p):
    d = len(p) - 1
    while d and p[d] == 0:
        d -= 1
    return d


def f12_inv(a):
    """extended Euclid over Fq[w]"""
    lm, hm = [1] + [0] * 12, [0] * 13
    low, high = list(a) + [0], [c % P for c in F12_MOD] + [1]
    while _poly_deg(low):
        # r = high / low (polynomial rounded division)
        dl, dh = _poly_deg(low), _poly_deg(high)
        temp, o = list(high), [0] * 13
        il = inv(low[dl], P)
        for i in range(dh - dl, -1, -1):
            o[i] = temp[dl + i] * il % P
            for c in range(dl + 1):
                temp[c + i] = (temp[c + i] - o[i] * low[c]) % P
        r = o
        nm, new = list(hm), list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] = (nm[i + j] - lm[i] * r[j]) % P
                new[i + j] = (new[i + j] - low[i] * r[j]) % P
        lm, low, hm, high = nm, new, lm, low
    il = inv(low[0], P)
    return [c * il % P for c in lm[:12]]


G12C = Curve(f12_add, f12_sub, f12_mul, lambda a: f12_mul(a, a), f12_inv, lambda a: [(-c) % P for c in a],
             f12([0]), F12_ONE, f12([3]), f12_scalar)
W = f12([0, 1])
W2, W3 = f12_mul(W, W), f12_mul(f12_mul(W, W), W)


def twist(pt):
    """G2 point over Fq2 -> point on y^2 = x^3 + 3 over Fq12."""
    if pt is None:
        return None
    (x0, x1), (y0, y1) = pt
    nx = f12([(x0 - 9 * x1) % P, 0, 0, 0, 0, 0, x1])
    ny = f12([(y0 - 9 * y1) % P, 0, 0, 0, 0, 0, y1])
    return (f12_mul(nx, W2), f12_mul(ny, W3))


def cast_g1(pt):
    return (f12([pt[0]]), f12([pt[1]]))


def linefunc(p1, p2, t):
    x1, y1 = p1
    x2, y2 = p2
    xt, yt = t
    if x1 != x2:
        m = f12_mul(f12_sub(y2, y1), f12_inv(f12_sub(x2, x1)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    if y1 == y2:
        m = f12_mul(f12_scalar(f12_mul(x1, x1), 3), f12_inv(f12_scalar(y1, 2)))
        return f12_sub(f12_mul(m, f12_sub(xt, x1)), f12_sub(yt, y1))
    return f12_sub(xt, x1)


def miller_loop(q, p):
    """q: twisted G2 point (Fq12 coordinates), p: G1 point cast to Fq12; no final exponentiation."""
    if q is None or p is None:
        return F12_ONE
    r, f = q, F12_ONE
    for i in range(LOG_ATE_LOOP_COUNT, -1, -1):
        f = f12_mul(f12_mul(f, f), linefunc(r, r, p))
        r = G12C.double(r)
        if ATE_LOOP_COUNT & (1 << i):
            f = f12_mul(f, linefunc(r, q, p))
            r = G12C.add_pts(r, q)
    q1 = (f12_pow(q[0], P), f12_pow(q[1], P))
    nq2 = (f12_pow(q1[0], P), [(-c) % P for c in f12_pow(q1[1], P)])
    f = f12_mul(f, linefunc(r, q1, p))
    r = G12C.add_pts(r, q1)
    f = f12_mul(f, linefunc(r, nq2, p))
    return f


FINAL_EXP = (P**12 - 1) // R


def final_exponentiate(f):
    return f12_pow(f, FINAL_EXP)


def pairing(q, p):
    return final_exponentiate(miller_loop(twist(q), cast_g1(p)))


def pairing_product_is_one(pairs):
    """prod e(p_i, q_i) == 1 with a single final exponentiation.  pairs = [(g1_point, g2_point), ...]"""
    f = F12_ONE
    for p, q in pairs:
        if p is None or q is None:
            continue
        f = f12_mul(f, miller_loop(twist(q), cast_g1(p)))
    return final_exponentiate(f) == F12_ONE


# ---------------------------------------------------------------- ark-serialize (uncompressed) of affine points
def _fq_bytes(x):
    return (x % P).to_bytes(32, "little")


def ser_g1(pt):
    """ark_ec short_weierstrass Affine::serialize_uncompressed: x || y, flags in the top bits of the last byte
    (bit 7: y is the lexicographically larger of {y, -y}; bit 6: point at infinity)."""
    if pt is None:
        out = bytearray(64)
        out[63] |= 0x40
        return bytes(out)
    out = bytearray(_fq_bytes(pt[0]) + _fq_bytes(pt[1]))
    if pt[1] > (-pt[1]) % P:
        out[63] |= 0x80
    return bytes(out)


def _f2_gt(a, b):       # ark_ff QuadExtField ordering: c1 first, then c0
    return (a[1], a[0]) > (b[1], b[0])


def ser_g2(pt):
    if pt is None:
        out = bytearray(128)
        out[127] |= 0x40
        return bytes(out)
    (x0, x1), (y0, y1) = pt
    out = bytearray(_fq_bytes(x0) + _fq_bytes(x1) + _fq_bytes(y0) + _fq_bytes(y1))
    if _f2_gt(pt[1], f2_neg(pt[1])):
        out[127] |= 0x80
    return bytes(out)


def de_g1(b):
    """ark-serialize, Compress::No, Validate::Yes: both flag bits set is an error (SWFlags::from_u8); the sign flag of a
    finite point is ignored (y comes from the bytes); canonical coordinates, on the curve (G1 has cofactor 1)."""
    if len(b) != 64:
        return False, None
    flags = b[63] & 0xC0
    if flags == 0xC0:
        return False, None
    if flags & 0x40:
        return True, None
    x = int.from_bytes(b[:32], "little")
    y = int.from_bytes(b[32:63] + bytes([b[63] & 0x3F]), "little")
    if x >= P or y >= P or not G1C.is_on_curve((x, y)):
        return False, None
    return True, (x, y)


def de_g2(b):
    if len(b) != 128:
        return False, None
    flags = b[127] & 0xC0
    if flags == 0xC0:
        return False, None
    if flags & 0x40:
        return True, None
    v = [int.from_bytes(b[32 * i: 32 * i + 32], "little") for i in range(3)]
    v.append(int.from_bytes(b[96:127] + bytes([b[127] & 0x3F]), "little"))
    if any(c >= P for c in v):
        return False, None
    pt = ((v[0], v[1]), (v[2], v[3]))
    if not G2C.is_on_curve(pt) or G2C.mul_pt(pt, R, reduce=False) is not None:     # subgroup check (G2 has a cofactor)
        return False, None
    return True, pt
