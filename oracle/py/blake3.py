"""ORACLE (test infrastructure only -- never imported by the product path).

BLAKE3 (default hash mode, 32-byte output) from the published specification, restricted to inputs of at most one
chunk (1024 bytes): everything the Winterfell pipeline of libzkp's improvement proof hashes is a row of field elements,
a pair of digests, a digest plus a counter, or a short seed (<= 176 bytes).  Stands in for the `blake3` crate behind
winterfell::crypto::hashers::Blake3_256 (/root/reference/src/backend/stark.rs:5,124).

Pinned by the specification's known answers for the empty input and "abc" (tests/test_oracle_stark.py).
"""
IV = (0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19)
PERM = (2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8)
CHUNK_START, CHUNK_END, PARENT, ROOT = 1, 2, 4, 8
M32 = 0xFFFFFFFF


def _rotr(x, n):
    return ((x >> n) | (x << (32 - n))) & M32


def _g(s, a, b, c, d, mx, my):
    s[a] = (s[a] + s[b] + mx) & M32
    s[d] = _rotr(s[d] ^ s[a], 16)
    s[c] = (s[c] + s[d]) & M32
    s[b] = _rotr(s[b] ^ s[c], 12)
    s[a] = (s[a] + s[b] + my) & M32
    s[d] = _rotr(s[d] ^ s[a], 8)
    s[c] = (s[c] + s[d]) & M32
    s[b] = _rotr(s[b] ^ s[c], 7)


def compress(cv, block_words, counter, block_len, flags):
    """the compression function; returns the 16-word state (first 8 words = next chaining value)"""
    s = list(cv) + list(IV[:4]) + [counter & M32, (counter >> 32) & M32, block_len, flags]
    m = list(block_words)
    for r in range(7):
        _g(s, 0, 4, 8, 12, m[0], m[1])
        _g(s, 1, 5, 9, 13, m[2], m[3])
        _g(s, 2, 6, 10, 14, m[4], m[5])
        _g(s, 3, 7, 11, 15, m[6], m[7])
        _g(s, 0, 5, 10, 15, m[8], m[9])
        _g(s, 1, 6, 11, 12, m[10], m[11])
        _g(s, 2, 7, 8, 13, m[12], m[13])
        _g(s, 3, 4, 9, 14, m[14], m[15])
        if r < 6:
            m = [m[PERM[i]] for i in range(16)]
    for i in range(8):
        s[i] ^= s[i + 8]
        s[i + 8] ^= cv[i]
    return s


def blake3(data):
    data = bytes(data)
    assert len(data) <= 1024, "single-chunk inputs only"
    blocks = [data[i:i + 64] for i in range(0, len(data), 64)] or [b""]
    cv = list(IV)
    for i, blk in enumerate(blocks):
        flags = (CHUNK_START if i == 0 else 0) | ((CHUNK_END | ROOT) if i == len(blocks) - 1 else 0)
        words = [int.from_bytes(blk.ljust(64, b"\0")[4 * k:4 * k + 4], "little") for k in range(16)]
        cv = compress(cv, words, 0, len(blk), flags)[:8]
    return b"".join(w.to_bytes(4, "little") for w in cv)
