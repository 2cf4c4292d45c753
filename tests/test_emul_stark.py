"""The STARK device header (libzkp_amd/csrc/stark_steps.h) compiled for the host against the oracle (no GPU):
f128 arithmetic, BLAKE3, the binding commitment and whole improvement-proof envelopes, bit for bit."""
import ctypes
import hashlib
import os
import random

import pytest

from oracle.py import stark as s
from oracle.py.blake3 import blake3


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build_emul()
    L = ctypes.CDLL(os.path.join(ge.EMUL_DIR, "_build", "libemul_stark.so"))
    L.emul_stark_prove.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint32]
    L.emul_improvement_commitment.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_void_p]
    return L


def test_f128_field(lib):
    rnd = random.Random(3)
    U2 = ctypes.c_uint64 * 2
    w = lambda x: U2(x & (2**64 - 1), x >> 64)  # noqa: E731
    out = U2()
    P = s.P
    vals = [0, 1, 2, P - 1, P - 2, 2**64 - 1, 2**64, 2**127, P >> 1, 2**128 - 2**46, P - 2**46] + [rnd.randrange(P) for _ in range(3000)]
    for i, a in enumerate(vals):
        b = vals[(7 * i + 3) % len(vals)] % P
        a %= P
        for op, f in ((0, a * b), (1, a + b), (2, a - b), (3, -a)):
            lib.emul_f128_op(op, w(a), w(b), out)
            assert out[0] | (out[1] << 64) == f % P, (op, a, b)
        if a and i < 40:
            lib.emul_f128_op(4, w(a), w(b), out)
            assert out[0] | (out[1] << 64) == pow(a, -1, P)


def test_blake3_and_commitment(lib):
    rnd = random.Random(5)
    for n in [0, 1, 4, 8, 10, 15, 16, 17, 32, 40, 64, 255, 256]:
        data = bytes(rnd.randrange(256) for _ in range(4 * n))
        arr = (ctypes.c_uint32 * max(n, 1))(*[int.from_bytes(data[4 * i:4 * i + 4], "little") for i in range(n)])
        o = (ctypes.c_uint32 * 8)()
        lib.emul_blake3_words(arr, n, o)
        assert b"".join(int(x).to_bytes(4, "little") for x in o) == blake3(data), n
    cm = (ctypes.c_uint8 * 32)()
    for old, new in ((0, 1), (2**64 - 2, 2**64 - 1), (123, 456)):
        lib.emul_improvement_commitment(old, new, cm)
        assert bytes(cm) == hashlib.sha256(b"libzkp_improvement_v1" + old.to_bytes(8, "little") + new.to_bytes(8, "little")).digest()


def test_envelopes_equal_oracle(lib):
    rnd = random.Random(6)
    cap = lib.emul_stark_max_envelope()
    assert cap == 3527
    buf = (ctypes.c_uint8 * cap)()
    cases = [(0, 1), (30, 50), (5, 2**63), (0, 2**64 - 1), (2**64 - 2, 2**64 - 1)]
    cases += [tuple(sorted((rnd.randrange(2**63), 2**63 + rnd.randrange(2**63)))) for _ in range(25)]
    cases += [(a, a + 1 + rnd.randrange(1000)) for a in (rnd.randrange(2**40) for _ in range(10))]
    lens = set()
    for old, new in cases:
        n = lib.emul_stark_prove(old, new, buf, cap)
        want = s.prove_improvement(old, new)
        assert bytes(buf[:n]) == want, (old, new)
        lens.add(n)
    assert len(lens) > 1          # the proof length really varies with the number of distinct query positions


def test_verifier_verdicts_equal_oracle(lib):
    """stark_verify_envelope (the GPU verifier's code, on the host) against oracle.verify_improvement: valid proofs, a wrong
    `old`, truncations, and a flipped bit at every ninth byte of several envelopes."""
    lib.emul_stark_verify.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint64]
    rnd = random.Random(5)
    n = 0
    for old, new in [(100, 250), (0, 1), (5, 2**63), (0, 2**64 - 1), (2**64 - 2, 2**64 - 1)]:
        env = s.prove_improvement(old, new)
        assert lib.emul_stark_verify(env, len(env), old) == 1
        assert lib.emul_stark_verify(env, len(env), old + 1) == 0
        for i in range(rnd.randrange(9), len(env), 9):
            bad = bytearray(env); bad[i] ^= 1 << rnd.randrange(8)
            assert bool(lib.emul_stark_verify(bytes(bad), len(bad), old)) == s.verify_improvement(bytes(bad), old), (old, new, i)
            n += 1
        for cut in (0, 1, 9, 10, 30, len(env) - 1):
            assert lib.emul_stark_verify(env[:cut], cut, old) == 0
    assert n > 1400
