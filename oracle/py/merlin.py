"""ORACLE (test infrastructure only -- never imported by the product path).

Keccak-f[1600], SHAKE256 / SHA3-512 (pinned against hashlib), STROBE-128 and the Merlin transcript.
The reference reaches this through the third-party crate `merlin ^3.0` (Cargo.toml:14; NOT vendored,
unpinned).  Call sites: /root/reference/src/backend/bulletproofs.rs:6,137,149,272,285,343,395,537,616,642,678
(`Transcript::new(b"libzkp_range_min")` etc.).  The published algorithm restated here is the Merlin
specification (merlin.cool, "Transcript protocol") over STROBE v1.0.2 with the 128-bit security
parameterisation (rate R = 166).  Known-answer pin: tests/test_oracle_merlin.py.
"""

MASK = (1 << 64) - 1
_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & MASK if n else x


def keccak_f1600(state: bytearray):
    a = [[int.from_bytes(state[8 * (x + 5 * y): 8 * (x + 5 * y) + 8], "little") for y in range(5)] for x in range(5)]
    for rnd in range(24):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= _RC[rnd]
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y): 8 * (x + 5 * y) + 8] = a[x][y].to_bytes(8, "little")


def _sponge(rate, suffix, data, outlen):
    st = bytearray(200)
    pos = 0
    for byte in data:
        st[pos] ^= byte
        pos += 1
        if pos == rate:
            keccak_f1600(st)
            pos = 0
    st[pos] ^= suffix
    st[rate - 1] ^= 0x80
    keccak_f1600(st)
    out = bytearray()
    while len(out) < outlen:
        out += st[:rate]
        if len(out) < outlen:
            keccak_f1600(st)
    return bytes(out[:outlen])


def shake256(data, outlen):
    return _sponge(136, 0x1F, data, outlen)


def sha3_512(data):
    return _sponge(72, 0x06, data, 64)


FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32
STROBE_R = 166


class Strobe128:
    def __init__(self, protocol_label: bytes):
        st = bytearray(200)
        st[0:6] = bytes([1, STROBE_R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f1600(st)
        self.state, self.pos, self.pos_begin, self.cur_flags = st, 0, 0, 0
        self.meta_ad(protocol_label, False)

    def clone(self):
        o = object.__new__(Strobe128)
        o.state, o.pos, o.pos_begin, o.cur_flags = bytearray(self.state), self.pos, self.pos_begin, self.cur_flags
        return o

    def _run_f(self):
        self.state[self.pos] ^= self.pos_begin
        self.state[self.pos + 1] ^= 0x04
        self.state[STROBE_R + 1] ^= 0x80
        keccak_f1600(self.state)
        self.pos = 0
        self.pos_begin = 0

    def _absorb(self, data):
        for byte in data:
            self.state[self.pos] ^= byte
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()

    def _squeeze(self, n):
        out = bytearray()
        for _ in range(n):
            out.append(self.state[self.pos])
            self.state[self.pos] = 0
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags, more):
        if more:
            assert self.cur_flags == flags
            return
        assert not (flags & FLAG_T)
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        if (flags & (FLAG_C | FLAG_K)) and self.pos != 0:
            self._run_f()

    def meta_ad(self, data, more):
        self._begin_op(FLAG_M | FLAG_A, more)
        self._absorb(data)

    def ad(self, data, more):
        self._begin_op(FLAG_A, more)
        self._absorb(data)

    def prf(self, n, more):
        self._begin_op(FLAG_I | FLAG_A | FLAG_C, more)
        return self._squeeze(n)


class Transcript:
    """merlin::Transcript."""

    def __init__(self, label: bytes):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label, message):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(len(message).to_bytes(4, "little"), True)
        self.strobe.ad(message, False)

    def append_u64(self, label, x):
        self.append_message(label, x.to_bytes(8, "little"))

    def challenge_bytes(self, label, n):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(n.to_bytes(4, "little"), True)
        return self.strobe.prf(n, False)
