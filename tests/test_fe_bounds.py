"""Interval-arithmetic proof that the 10x25.5-limb field code (libzkp_amd/csrc/fe25519.h) never overflows
its 32-bit operands or 64-bit column sums for the limb bounds its comments promise (no GPU)."""

EVEN, ODD = 0, 1


def carried():
    return [2**26 if i % 2 == 0 else 2**25 + 2**18 for i in range(10)]     # exclusive upper bounds


def add(f, g):
    return [a + b for a, b in zip(f, g)]


def sub(f, g_carried):
    two_p = [0x7FFFFDA] + [0x3FFFFFE if i % 2 else 0x7FFFFFE for i in range(1, 10)]
    assert all(gb - 1 <= t for gb, t in zip(g_carried, two_p)), "fe_sub subtrahend must be carried"
    return [a + t + 1 for a, t in zip(f, two_p)]


def mul(f, g):
    """returns carried bound; asserts no overflow for exclusive upper bounds f, g"""
    assert all(19 * (x - 1) < 2**32 for x in g), "19*g must fit in 32 bits"
    assert all(2 * (x - 1) < 2**32 for x in f)
    for k in range(10):
        acc = 0
        for i in range(10):
            j = (k - i) % 10
            wrap = (k - i) < 0
            fi = (f[i] - 1) * (2 if (i % 2 and j % 2) else 1)
            acc += fi * ((g[j] - 1) * (19 if wrap else 1))
        # the carry chain adds at most 2^39 from the previous column
        assert acc + 2**40 < 2**64, ("column", k, acc.bit_length())
    return carried()


def sq(f):
    assert all(19 * (x - 1) < 2**32 and 4 * (x - 1) < 2**32 for x in f)
    return mul(f, f)


def finish_add(A, B, C, D):
    E, F, G, H = sub(B, A), sub(D, C), add(D, C), add(B, A)
    return mul(F, E), mul(G, H), mul(F, G), mul(E, H)


def test_madd_bounds():
    c = carried()
    A = mul(sub(c, c), c)          # (Y1-X1) * ymx
    B = mul(add(c, c), c)          # (Y1+X1) * ypx
    C = mul(c, c)
    negated = [0x7FFFFDA + 1] + [(0x3FFFFFE if i % 2 else 0x7FFFFFE) + 1 for i in range(1, 10)]   # 2p - xy2d
    mul(c, negated)
    D = add(c, c)
    finish_add(A, B, C, D)


def test_add_and_double_bounds():
    c = carried()
    A = mul(sub(c, c), sub(c, c))
    B = mul(add(c, c), add(c, c))
    C = mul(mul(c, c), c)
    zz = mul(c, c)
    finish_add(A, B, C, add(zz, zz))
    # ge_dbl
    a, b, z2 = sq(c), sq(c), sq(c)
    Cc = add(z2, z2)
    t = sq(add(c, c))
    Hn = add(a, b)
    En = sub(Hn, t)
    Gn = sub(a, b)
    Fn = carried()                 # fe_carry(C + Gn)
    assert all(x + y < 2**32 for x, y in zip(Cc, Gn))
    mul(En, Fn), mul(Gn, Hn), mul(Fn, Gn), mul(En, Hn)


def test_loose_times_semi_loose_is_the_documented_limit():
    loose = [2**28 if i % 2 == 0 else 2**27 for i in range(10)]
    semi = [3 * 2**26 if i % 2 == 0 else 3 * 2**25 for i in range(10)]
    mul(loose, semi)
    sq(semi)
