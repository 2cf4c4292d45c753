"""ORACLE (test infrastructure only -- never imported by the product path).

Bigint restatement of the group arithmetic that the reference reaches through the third-party
crate `curve25519-dalek ^4.1` (Cargo.toml:13; NOT vendored under /root/reference, version unpinned:
no Cargo.lock).  Call sites that fix the semantics we need:
  /root/reference/src/backend/bulletproofs.rs:4-5   RistrettoPoint / CompressedRistretto / Scalar
  /root/reference/src/backend/bulletproofs.rs:86    Scalar::from_bytes_mod_order
  /root/reference/src/backend/bulletproofs.rs:134   pc_gens.commit(..).compress()
  /root/reference/src/backend/bulletproofs.rs:191   CompressedRistretto::decompress
The published algorithm restated here is RFC 9496 (ristretto255) over edwards25519 (RFC 8032).
Pinned in tests/test_oracle_ristretto.py against fixtures generated with libsodium 1.0.18
(tests/golden/ristretto_libsodium.json, generator script tests/golden/gen_ristretto_libsodium.py).
"""

P = 2**255 - 19
L = 2**252 + 27742317777372353535851937790883648493
D = (-121665 * pow(121666, P - 2, P)) % P
SQRT_M1 = pow(2, (P - 1) // 4, P)


def _is_neg(x):
    return (x % P) & 1


def _abs(x):
    x %= P
    return P - x if x & 1 else x


def sqrt_ratio_m1(u, v):
    """RFC 9496 section 4.2 SQRT_RATIO_M1 -> (was_square, r)."""
    u %= P
    v %= P
    v3 = v * v % P * v % P
    v7 = v3 * v3 % P * v % P
    r = (u * v3 % P) * pow(u * v7 % P, (P - 5) // 8, P) % P
    check = v * r % P * r % P
    correct = check == u
    flipped = check == (-u) % P
    flipped_i = check == (-u * SQRT_M1) % P
    if flipped or flipped_i:
        r = r * SQRT_M1 % P
    return (correct or flipped), _abs(r)


INVSQRT_A_MINUS_D = sqrt_ratio_m1(1, (-1 - D) % P)[1]
ONE_MINUS_D_SQ = (1 - D * D) % P
D_MINUS_ONE_SQ = (D - 1) * (D - 1) % P
# sqrt(a*d - 1) with the sign RFC 9496 fixes (the published constant is the odd root's negation choice:
# 25063068953384623474111414158702152701244531502492656460079210482610430750235)
SQRT_AD_MINUS_ONE = 25063068953384623474111414158702152701244531502492656460079210482610430750235
assert SQRT_AD_MINUS_ONE * SQRT_AD_MINUS_ONE % P == (-D - 1) % P
assert SQRT_M1 == 19681161376707505956807079304988542015446066515923890162744021073123829784752
assert INVSQRT_A_MINUS_D == 54469307008909316920995813868745141605393597292927456921205312896311721017578


class Point:
    """edwards25519 point, extended coordinates (X:Y:Z:T), a = -1."""

    __slots__ = ("X", "Y", "Z", "T")

    def __init__(self, X, Y, Z, T):
        self.X, self.Y, self.Z, self.T = X % P, Y % P, Z % P, T % P

    def __add__(self, o):
        # add-2008-hwcd-3 (a=-1), complete for our (odd-order-coset) inputs
        A = (self.Y - self.X) * (o.Y - o.X) % P
        B = (self.Y + self.X) * (o.Y + o.X) % P
        C = self.T * 2 * D % P * o.T % P
        Dd = self.Z * 2 * o.Z % P
        E, F, G, H = B - A, Dd - C, Dd + C, B + A
        return Point(E * F, G * H, F * G, E * H)

    def __neg__(self):
        return Point(-self.X, self.Y, self.Z, -self.T)

    def __sub__(self, o):
        return self + (-o)

    def double(self):
        A = self.X * self.X % P
        B = self.Y * self.Y % P
        C = 2 * self.Z * self.Z % P
        Dd = -A % P
        E = ((self.X + self.Y) ** 2 - A - B) % P
        G = Dd + B
        F = G - C
        H = Dd - B
        return Point(E * F, G * H, F * G, E * H)

    def __rmul__(self, k):
        k %= L
        acc = IDENTITY
        # fixed 4-bit window (oracle speed only)
        tbl = [IDENTITY, self]
        for _ in range(14):
            tbl.append(tbl[-1] + self)
        for shift in range(252, -1, -4):
            if shift != 252:
                acc = acc.double().double().double().double()
            nib = (k >> shift) & 15
            if nib:
                acc = acc + tbl[nib]
        return acc

    def __eq__(self, o):
        # ristretto equality (RFC 9496 4.3.3... section 4.5): X1*Y2 == Y1*X2 or Y1*Y2 == X1*X2
        return (self.X * o.Y - self.Y * o.X) % P == 0 or (self.Y * o.Y - self.X * o.X) % P == 0

    def is_identity(self):
        return self == IDENTITY

    def encode(self):
        """RFC 9496 section 4.3.2 Encode."""
        X, Y, Z, T = self.X, self.Y, self.Z, self.T
        u1 = (Z + Y) * (Z - Y) % P
        u2 = X * Y % P
        _, invsqrt = sqrt_ratio_m1(1, u1 * u2 % P * u2 % P)
        den1 = invsqrt * u1 % P
        den2 = invsqrt * u2 % P
        z_inv = den1 * den2 % P * T % P
        ix = X * SQRT_M1 % P
        iy = Y * SQRT_M1 % P
        ench = den1 * INVSQRT_A_MINUS_D % P
        if _is_neg(T * z_inv):
            x, y, den_inv = iy, ix, ench
        else:
            x, y, den_inv = X, Y, den2
        if _is_neg(x * z_inv):
            y = -y % P
        s = _abs(den_inv * (Z - y))
        return s.to_bytes(32, "little")


IDENTITY = Point(0, 1, 1, 0)
_BY = 4 * pow(5, P - 2, P) % P
_BX = 15112221349535400772501151409588531511454012693041857206046113283949847762202
BASEPOINT = Point(_BX, _BY, 1, _BX * _BY)
assert (-_BX * _BX + _BY * _BY - 1 - D * _BX * _BX % P * _BY * _BY) % P == 0


def decode(b):
    """RFC 9496 section 4.3.1 Decode -> Point or None."""
    if len(b) != 32:
        return None
    s = int.from_bytes(b, "little")
    if s >= P or (s & 1):
        return None
    ss = s * s % P
    u1 = (1 - ss) % P
    u2 = (1 + ss) % P
    u2_sqr = u2 * u2 % P
    v = (-(D * u1 % P * u1) - u2_sqr) % P
    was_square, invsqrt = sqrt_ratio_m1(1, v * u2_sqr % P)
    den_x = invsqrt * u2 % P
    den_y = invsqrt * den_x % P * v % P
    x = _abs(2 * s * den_x)
    y = u1 * den_y % P
    t = x * y % P
    if (not was_square) or _is_neg(t) or y == 0:
        return None
    return Point(x, y, 1, t)


def _map(t):
    """RFC 9496 section 4.3.4 MAP (Elligator 2 on the Jacobi quartic)."""
    r = SQRT_M1 * t % P * t % P
    u = (r + 1) * ONE_MINUS_D_SQ % P
    v = (-1 - r * D) % P * ((r + D) % P) % P
    was_square, s = sqrt_ratio_m1(u, v)
    s_prime = -_abs(s * t) % P
    if not was_square:
        s = s_prime
        c = r
    else:
        c = P - 1
    N = (c * (r - 1) % P * D_MINUS_ONE_SQ - v) % P
    w0 = 2 * s * v % P
    w1 = N * SQRT_AD_MINUS_ONE % P
    w2 = (1 - s * s) % P
    w3 = (1 + s * s) % P
    return Point(w0 * w3, w2 * w1, w1 * w3, w0 * w2)


def from_uniform_bytes(b):
    """RistrettoPoint::from_uniform_bytes == RFC 9496 one-way map (== libsodium ..._from_hash)."""
    assert len(b) == 64
    t1 = int.from_bytes(b[:32], "little") & ((1 << 255) - 1)
    t2 = int.from_bytes(b[32:], "little") & ((1 << 255) - 1)
    return _map(t1 % P) + _map(t2 % P)


def scalar_from_bytes_mod_order(b):
    return int.from_bytes(b, "little") % L


def scalar_to_bytes(s):
    return (s % L).to_bytes(32, "little")


def scalar_from_canonical_bytes(b):
    s = int.from_bytes(b, "little")
    return s if s < L else None


def msm(scalars, points):
    acc = IDENTITY
    for k, pt in zip(scalars, points):
        k %= L
        if k:
            acc = acc + k * pt
    return acc
