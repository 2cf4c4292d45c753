"""BN254 device headers and the Groth16 witness/QAP steps compiled for the host, against the oracle (no GPU)."""
import ctypes
import os
import random

import numpy as np
import pytest

from oracle.py import bn254 as b
from oracle.py import groth16 as g


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
    pts2 = (ctypes.c_uint32 * (32 * n))(*sum((g2w(b.G2C.mul_pt(b.G2, k)) for k in ks), []))
    lib.emul_g2_lazy_chain((ctypes.c_uint32 * 32)(*g2w(b.G2C.mul_pt(b.G2, ko))), pts2, arr, n, o32)
    assert I(o32).to_bytes(128, "little") == b.ser_g2(b.G2C.mul_pt(b.G2, total))


def _run(lib, kind, value, the_set, seed):
    nv, m = (334, 512) if kind == 0 else (653, 1024)
    z = np.zeros((nv, 8), dtype=np.uint32)
    h = np.zeros((m - 1, 8), dtype=np.uint32)
    rs = np.zeros(16, dtype=np.uint32)
    out = np.zeros(1024, dtype=np.uint8)
    shape = np.zeros(4, dtype=np.uint32)
    sv = np.array(list(the_set) + [0], dtype=np.uint64)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    lib.emul_g16_witness_qap(kind, ctypes.c_uint64(value), P(sv), len(the_set), seed, P(z), P(h), P(rs), P(out), ctypes.c_uint64(1024), P(shape))
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
