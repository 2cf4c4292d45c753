// libzkp's improvement proof (Winterfell STARK, 8-row x 1-column trace over p = 2^128 - 45*2^40 + 1) as cooperative
// per-proof steps: `nthreads` lanes (64 on the GPU = one wavefront per proof, 1 in host emulation) share a ProofMem
// block (LDS on the GPU); `sync` is the barrier between steps.  Lane i owns row i of the 64-point LDE domain: it
// evaluates the trace / composition column there, hashes the row (BLAKE3) and takes part in the Merkle levels; the
// Fiat-Shamir coin, the out-of-domain frame, the DEEP/FRI remainder and the serialisation run on lane 0 (they are a
// strictly serial hash chain of ~25 compressions).
// Replaces winterfell's Prover::prove as called by StarkBackend::prove_improvement (/root/reference/src/backend/stark.rs:151-186)
// and the envelope framing of proof/improvement_proof.rs:10-35 + utils/commitment.rs:38-50.  Pipeline and byte layout:
// see oracle/py/stark.py (the restatement this code is tested against, bit for bit).
#pragma once
#include "zkp_common.h"
#include "sha256_dev.h"
// The hash compression, the field product and the serial tail are real (non-inlined) device functions: the kernel is
// latency-bound on lane 0's hash chain, not on call overhead, and full inlining made a 60k-instruction kernel that took
// 12 minutes to compile.

namespace zkp {

// ---------------------------------------------------------------------------------------------- field
struct f128 { uint64_t lo, hi; };
#define ZKP_F128_C 0x2CFFFFFFFFFFull            /* 2^128 mod p = 45 * 2^40 - 1 */
#define ZKP_F128_PLO 0xFFFFD30000000001ull      /* p = 2^128 - 45 * 2^40 + 1 */
#define ZKP_F128_PHI 0xFFFFFFFFFFFFFFFFull

ZKP_HD inline void mul64wide(uint64_t a, uint64_t b, uint64_t& hi, uint64_t& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    lo = a * b; hi = __umul64hi(a, b);
#else
    const unsigned __int128 p = (unsigned __int128)a * b; lo = (uint64_t)p; hi = (uint64_t)(p >> 64);
#endif
}
ZKP_HD inline f128 f128_make(uint64_t lo, uint64_t hi = 0) { f128 r; r.lo = lo; r.hi = hi; return r; }
ZKP_HD inline bool f128_geq_p(const f128& a) { return a.hi == ZKP_F128_PHI && a.lo >= ZKP_F128_PLO; }
ZKP_HD inline f128 f128_canon(const f128& a) { return f128_geq_p(a) ? f128_make(a.lo - ZKP_F128_PLO, 0) : a; }
// all routines take and return canonical values (< p)
ZKP_HD inline f128 f128_add(const f128& a, const f128& b) {
    f128 r; r.lo = a.lo + b.lo; const uint64_t c0 = r.lo < a.lo;
    r.hi = a.hi + b.hi; const uint64_t c1 = r.hi < a.hi; r.hi += c0; const uint64_t c2 = r.hi < c0;
    if (c1 | c2) {                                  // wrapped 2^128: add 2^128 mod p (cannot wrap again: the sum is < 2p)
        const uint64_t t = r.lo + ZKP_F128_C; r.hi += t < r.lo; r.lo = t;
    }
    return f128_canon(r);
}
ZKP_HD inline f128 f128_neg(const f128& a) {
    if ((a.lo | a.hi) == 0) return a;
    f128 r; r.lo = ZKP_F128_PLO - a.lo; r.hi = ZKP_F128_PHI - a.hi - (ZKP_F128_PLO < a.lo); return r;
}
ZKP_HD inline f128 f128_sub(const f128& a, const f128& b) { return f128_add(a, f128_neg(b)); }
ZKP_HD_NOINLINE inline f128 f128_mul(const f128& a, const f128& b) {
    // 256-bit product w3..w0
    uint64_t w0, w1, w2, w3, h, l;
    mul64wide(a.lo, b.lo, w1, w0);
    mul64wide(a.hi, b.hi, w3, w2);
    mul64wide(a.lo, b.hi, h, l);
    w1 += l; uint64_t c = w1 < l; w2 += c; c = w2 < c; w3 += c;
    w2 += h; w3 += w2 < h;
    mul64wide(a.hi, b.lo, h, l);
    w1 += l; c = w1 < l; w2 += c; c = w2 < c; w3 += c;
    w2 += h; w3 += w2 < h;
    // (w3 w2) * C : 128 x 46 bits -> 174 bits (t2 t1 t0), t2 < 2^46
    uint64_t t0, t1, t2, x;
    mul64wide(w2, ZKP_F128_C, t1, t0);
    mul64wide(w3, ZKP_F128_C, t2, x);
    t1 += x; t2 += t1 < x;
    // t2 * C < 2^92
    uint64_t u0, u1;
    mul64wide(t2, ZKP_F128_C, u1, u0);
    // r = (w1 w0) + (t1 t0) + (u1 u0), counting the wraps past 2^128 (at most 2)
    f128 r; uint64_t wraps = 0;
    r.lo = w0 + t0; c = r.lo < w0;
    r.hi = w1 + t1; wraps += r.hi < w1; r.hi += c; wraps += r.hi < c;
    x = r.lo + u0; c = x < r.lo; r.lo = x;
    x = r.hi + u1; wraps += x < r.hi; r.hi = x; r.hi += c; wraps += r.hi < c;
    while (wraps) {                                  // each wrap past 2^128 is worth C; adding it may (rarely) wrap once more
        wraps--;
        x = r.lo + ZKP_F128_C; c = x < r.lo; r.lo = x;
        x = r.hi + c; if (x < r.hi) wraps++; r.hi = x;
    }
    return f128_canon(r);
}
ZKP_HD inline f128 f128_pow(const f128& a, uint64_t e_lo, uint64_t e_hi) {
    f128 acc = f128_make(1);
    for (int i = 127; i >= 0; i--) {
        acc = f128_mul(acc, acc);
        const uint64_t bit = i >= 64 ? (e_hi >> (i - 64)) & 1 : (e_lo >> i) & 1;
        if (bit) acc = f128_mul(acc, a);
    }
    return acc;
}
ZKP_HD inline f128 f128_inv(const f128& a) { return f128_pow(a, ZKP_F128_PLO - 2, ZKP_F128_PHI); }   // host-side table construction only

// ---------------------------------------------------------------------------------------------- BLAKE3 (inputs <= 1 chunk, whole words)
ZKP_HD constexpr uint32_t blake3_iv(int i) { constexpr uint32_t iv[8] = {0x6A09E667u, 0xBB67AE85u, 0x3C6EF372u, 0xA54FF53Au, 0x510E527Fu, 0x9B05688Cu, 0x1F83D9ABu, 0x5BE0CD19u}; return iv[i]; }
ZKP_HD inline uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
#define ZKP_B3_G(a, b, c, d, mx, my)                                                                      \
    do {                                                                                                  \
        s[a] = s[a] + s[b] + (mx); s[d] = rotr32(s[d] ^ s[a], 16); s[c] = s[c] + s[d]; s[b] = rotr32(s[b] ^ s[c], 12); \
        s[a] = s[a] + s[b] + (my); s[d] = rotr32(s[d] ^ s[a], 8); s[c] = s[c] + s[d]; s[b] = rotr32(s[b] ^ s[c], 7);   \
    } while (0)
// chaining value update for one block (counter 0)
ZKP_HD_NOINLINE inline void blake3_compress(uint32_t cv[8], const uint32_t block[16], uint32_t block_len, uint32_t flags) {
    uint32_t s[16], m[16];
    ZKP_UNROLL for (int i = 0; i < 8; i++) s[i] = cv[i];
    ZKP_UNROLL for (int i = 0; i < 4; i++) s[8 + i] = blake3_iv(i);
    s[12] = 0; s[13] = 0; s[14] = block_len; s[15] = flags;
    ZKP_UNROLL for (int i = 0; i < 16; i++) m[i] = block[i];
    ZKP_UNROLL for (int r = 0; r < 7; r++) {
        ZKP_B3_G(0, 4, 8, 12, m[0], m[1]); ZKP_B3_G(1, 5, 9, 13, m[2], m[3]); ZKP_B3_G(2, 6, 10, 14, m[4], m[5]); ZKP_B3_G(3, 7, 11, 15, m[6], m[7]);
        ZKP_B3_G(0, 5, 10, 15, m[8], m[9]); ZKP_B3_G(1, 6, 11, 12, m[10], m[11]); ZKP_B3_G(2, 7, 8, 13, m[12], m[13]); ZKP_B3_G(3, 4, 9, 14, m[14], m[15]);
        if (r < 6) {
            const uint32_t t[16] = {m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8]};
            ZKP_UNROLL for (int i = 0; i < 16; i++) m[i] = t[i];
        }
    }
    ZKP_UNROLL for (int i = 0; i < 8; i++) cv[i] = s[i] ^ s[i + 8];
}
// BLAKE3 of nwords 32-bit little-endian words (nwords <= 256)
ZKP_HD_NOINLINE inline void blake3_words(uint32_t out[8], const uint32_t* in, uint32_t nwords) {
    ZKP_UNROLL for (int i = 0; i < 8; i++) out[i] = blake3_iv(i);
    const uint32_t nblocks = nwords == 0 ? 1 : (nwords + 15) / 16;
    for (uint32_t b = 0; b < nblocks; b++) {
        uint32_t m[16];
        const uint32_t have = nwords - 16 * b < 16 ? nwords - 16 * b : 16;
        ZKP_UNROLL for (uint32_t i = 0; i < 16; i++) m[i] = i < have ? in[16 * b + i] : 0u;
        blake3_compress(out, m, 4 * have, (b == 0 ? 1u : 0u) | (b == nblocks - 1 ? 10u : 0u));   // CHUNK_START | CHUNK_END + ROOT
    }
}
ZKP_HD inline void blake3_merge(uint32_t out[8], const uint32_t a[8], const uint32_t b[8]) {
    uint32_t m[16]; ZKP_UNROLL for (int i = 0; i < 8; i++) { m[i] = a[i]; m[8 + i] = b[i]; }
    ZKP_UNROLL for (int i = 0; i < 8; i++) out[i] = blake3_iv(i);
    blake3_compress(out, m, 64, 11);
}
ZKP_HD inline void blake3_merge_int(uint32_t out[8], const uint32_t seed[8], uint64_t v) {
    uint32_t m[16]; ZKP_UNROLL for (int i = 0; i < 8; i++) { m[i] = seed[i]; m[8 + i] = 0; }
    m[8] = (uint32_t)v; m[9] = (uint32_t)(v >> 32);
    ZKP_UNROLL for (int i = 0; i < 8; i++) out[i] = blake3_iv(i);
    blake3_compress(out, m, 40, 11);
}
ZKP_HD inline void f128_words(uint32_t w[4], const f128& a) { w[0] = (uint32_t)a.lo; w[1] = (uint32_t)(a.lo >> 32); w[2] = (uint32_t)a.hi; w[3] = (uint32_t)(a.hi >> 32); }
ZKP_HD inline void blake3_elements(uint32_t out[8], const f128* e, uint32_t n) {   // n <= 16
    uint32_t w[64];
    for (uint32_t i = 0; i < n; i++) f128_words(w + 4 * i, e[i]);
    blake3_words(out, w, 4 * n);
}

// ---------------------------------------------------------------------------------------------- the binding commitment
// SHA-256("libzkp_improvement_v1" || u64le(old) || u64le(new))  (utils/commitment.rs:38-50): 37 bytes, one padded block, built in
// registers -- bytes 0..20 the tag, 21..28 old, 29..36 new (little-endian), 0x80, zeros, the bit length 296 in the last word
ZKP_HD inline void improvement_commitment(uint8_t out[32], uint64_t oldv, uint64_t newv) {
    auto byte_at = [&](int i) -> uint32_t {
        const char tag[22] = "libzkp_improvement_v1";
        return i < 21 ? (uint32_t)(uint8_t)tag[i] : i < 29 ? (uint32_t)(oldv >> (8 * (i - 21))) & 0xffu : i < 37 ? (uint32_t)(newv >> (8 * (i - 29))) & 0xffu : i == 37 ? 0x80u : 0u;
    };
    uint32_t m[16], h[8];
    ZKP_UNROLL for (int k = 0; k < 16; k++) m[k] = (byte_at(4 * k) << 24) | (byte_at(4 * k + 1) << 16) | (byte_at(4 * k + 2) << 8) | byte_at(4 * k + 3);
    m[15] = 37 * 8;
    sha256_one_block_words(h, m);
    for (int k = 0; k < 8; k++) { out[4 * k] = (uint8_t)(h[k] >> 24); out[4 * k + 1] = (uint8_t)(h[k] >> 16); out[4 * k + 2] = (uint8_t)(h[k] >> 8); out[4 * k + 3] = (uint8_t)h[k]; }
}

// ---------------------------------------------------------------------------------------------- the prover
constexpr uint32_t STARK_TRACE_LEN = 8, STARK_LDE = 64, STARK_CE = 16, STARK_QUERIES = 32, STARK_DEPTH = 6;
constexpr uint32_t STARK_MAX_PROOF = 3469;                    // 32 distinct positions, 32 sibling nodes per opening
constexpr uint32_t STARK_MAX_ENVELOPE = 10 + 16 + STARK_MAX_PROOF + 32;

// domain constants, built once on the host (stark_build_constants) and kept in device memory
struct StarkConst {
    f128 x_lde[STARK_LDE];              // 3 * w64^i
    f128 w8_inv_pow[8], w16_inv_pow[16];
    f128 scale8;                        // 1/8
    f128 scale16[8];                    // 3^-k / 16
    f128 inv_zt[STARK_CE], inv_xm1[STARK_CE], inv_xml[STARK_CE];   // 1 / Z_t(x), 1/(x - 1), 1/(x - g^7) on the 16-point coset
    f128 inv7, g, g_last;               // 1/7, w8, w8^7
    uint32_t seed_prefix[32];           // the eight context elements of the coin seed, as words
    uint8_t context_bytes[28];
};
inline void stark_build_constants(StarkConst& C) {
    const f128 gen = f128_make(3);
    // two-adic root: 3^((p-1)/2^40); (p-1)/2^40 = 2^88 - 45
    const f128 tw = f128_pow(gen, (uint64_t)0 - 45ull, (1ull << 24) - 1);   // 2^88 - 45 = (2^24 - 1) * 2^64 + (2^64 - 45)
    auto root = [&](uint32_t log_n) { f128 r = tw; for (uint32_t i = log_n; i < 40; i++) r = f128_mul(r, r); return r; };
    const f128 w64 = root(6), w16 = root(4), w8 = root(3);
    f128 x = gen;
    for (uint32_t i = 0; i < STARK_LDE; i++) { C.x_lde[i] = x; x = f128_mul(x, w64); }
    const f128 w8i = f128_inv(w8), w16i = f128_inv(w16);
    C.w8_inv_pow[0] = f128_make(1); for (int i = 1; i < 8; i++) C.w8_inv_pow[i] = f128_mul(C.w8_inv_pow[i - 1], w8i);
    C.w16_inv_pow[0] = f128_make(1); for (int i = 1; i < 16; i++) C.w16_inv_pow[i] = f128_mul(C.w16_inv_pow[i - 1], w16i);
    C.scale8 = f128_inv(f128_make(8));
    const f128 gi = f128_inv(gen); f128 s = f128_inv(f128_make(16));
    for (int k = 0; k < 8; k++) { C.scale16[k] = s; s = f128_mul(s, gi); }
    C.inv7 = f128_inv(f128_make(7)); C.g = w8;
    f128 last = f128_make(1); for (int i = 0; i < 7; i++) last = f128_mul(last, w8);
    C.g_last = last;
    for (uint32_t i = 0; i < STARK_CE; i++) {
        const f128 xc = C.x_lde[i * (STARK_LDE / STARK_CE)];
        f128 x8 = xc; for (int k = 0; k < 3; k++) x8 = f128_mul(x8, x8);
        const f128 zt = f128_mul(f128_sub(x8, f128_make(1)), f128_inv(f128_sub(xc, last)));
        C.inv_zt[i] = f128_inv(zt); C.inv_xm1[i] = f128_inv(f128_sub(xc, f128_make(1))); C.inv_xml[i] = f128_inv(f128_sub(xc, last));
    }
    const f128 ctx[8] = {f128_make(256), f128_make(8), f128_make(ZKP_F128_PLO), f128_make(ZKP_F128_PHI), f128_make((1u << 16) | (8u << 8) | 31u), f128_make(0), f128_make(8), f128_make(32)};
    for (int i = 0; i < 8; i++) f128_words(C.seed_prefix + 4 * i, ctx[i]);
    const uint8_t head[6] = {1, 0, 3, 0, 0, 16};
    for (int i = 0; i < 6; i++) C.context_bytes[i] = head[i];
    for (int i = 0; i < 8; i++) { C.context_bytes[6 + i] = (uint8_t)(ZKP_F128_PLO >> (8 * i)); C.context_bytes[14 + i] = 0xFF; }
    const uint8_t opt[6] = {32, 8, 0, 1, 8, 31};
    for (int i = 0; i < 6; i++) C.context_bytes[22 + i] = opt[i];
}

// per-proof shared block (LDS on the GPU)
// Round 4: nothing of the prover lives in private (scratch) memory any more.  Round 3's kernel passed digests and message blocks to the
// non-inlined compression function as pointers to per-lane arrays, which puts those arrays in scratch: 976 bytes per lane, and every
// one of lane 0's ~45 serial compressions paid several round trips to it (VALU active 19 % of the wave's life).  Now a lane hands its
// chaining value and message block to the ONE copy of the compression through its column of `hio` (word-major: word w of lane l at
// hio[w * 64 + l], conflict-free), the bookkeeping of the batch Merkle openings sits in `plan_*`, and the three loops that were serial
// on lane 0 without needing to be -- the 32 query-position draws (32 independent compressions), the copy of ~100 digests and ~60 field
// elements into the envelope -- are spread over the wave.  What remains serial is the Fiat-Shamir chain itself (~14 compressions).
constexpr uint32_t STARK_HIO_STRIDE = 64, STARK_PLAN_MAX = 192;
struct StarkMem {
    f128 t_poly[8], h_poly[8], t_lde[STARK_LDE], h_lde[STARK_LDE], ce[STARK_CE], col[8];
    uint32_t t_leaf[STARK_LDE][8], h_leaf[STARK_LDE][8], t_node[STARK_LDE][8], h_node[STARK_LDE][8];   // node i: children 2i, 2i+1; [1] = root
    uint32_t seed[8]; uint64_t counter;
    f128 coef[3], step, oldv, newv;
    uint32_t hio[24 * STARK_HIO_STRIDE];                          // hash I/O columns: words 0..7 chaining value / digest, 8..23 message block
    f128 ood[3], rem[8]; uint32_t rc[8];                          // T(z), T(z g), H(z); the DEEP remainder and its commitment
    uint64_t mask; uint32_t count;                                // queried positions
    uint8_t plan_code[STARK_PLAN_MAX]; uint16_t plan_off[STARK_PLAN_MAX];     // opening digests: (0x80 | leaf) or node index, byte offset inside an opening
    uint32_t plan_n, lists_off[2], val_off[2], tail_off;          // digests per opening; where each opening's node lists, each query block's values and the tail start
    uint8_t w_idx[32], w_nxt[32], w_cnt[32], w_store[32][6];      // lane 0's bookkeeping of the batch opening
    uint32_t out_len;
    uint8_t out[STARK_MAX_ENVELOPE + 3];
};

// ---- BLAKE3 through a lane's hash I/O column (io = M.hio + lane)
ZKP_HD_NOINLINE inline void blake3_compress_io(uint32_t* io, uint32_t block_len, uint32_t flags) {
    uint32_t s[16], m[16];
    ZKP_UNROLL for (int i = 0; i < 8; i++) s[i] = io[i * STARK_HIO_STRIDE];
    ZKP_UNROLL for (int i = 0; i < 4; i++) s[8 + i] = blake3_iv(i);
    s[12] = 0; s[13] = 0; s[14] = block_len; s[15] = flags;
    ZKP_UNROLL for (int i = 0; i < 16; i++) m[i] = io[(8 + i) * STARK_HIO_STRIDE];
    ZKP_UNROLL for (int r = 0; r < 7; r++) {
        ZKP_B3_G(0, 4, 8, 12, m[0], m[1]); ZKP_B3_G(1, 5, 9, 13, m[2], m[3]); ZKP_B3_G(2, 6, 10, 14, m[4], m[5]); ZKP_B3_G(3, 7, 11, 15, m[6], m[7]);
        ZKP_B3_G(0, 5, 10, 15, m[8], m[9]); ZKP_B3_G(1, 6, 11, 12, m[10], m[11]); ZKP_B3_G(2, 7, 8, 13, m[12], m[13]); ZKP_B3_G(3, 4, 9, 14, m[14], m[15]);
        if (r < 6) {
            const uint32_t t[16] = {m[2], m[6], m[3], m[10], m[7], m[0], m[4], m[13], m[1], m[11], m[12], m[5], m[9], m[14], m[15], m[8]};
            ZKP_UNROLL for (int i = 0; i < 16; i++) m[i] = t[i];
        }
    }
    ZKP_UNROLL for (int i = 0; i < 8; i++) io[i * STARK_HIO_STRIDE] = s[i] ^ s[i + 8];
}
ZKP_HD inline void hio_iv(uint32_t* io) { ZKP_UNROLL for (int i = 0; i < 8; i++) io[i * STARK_HIO_STRIDE] = blake3_iv(i); }
ZKP_HD inline void hio_msg(uint32_t* io, int i, uint32_t v) { io[(8 + i) * STARK_HIO_STRIDE] = v; }
ZKP_HD inline void hio_msg_zero(uint32_t* io, int from) { for (int i = from; i < 16; i++) hio_msg(io, i, 0u); }
ZKP_HD inline void hio_digest(uint32_t out[8], const uint32_t* io) { ZKP_UNROLL for (int i = 0; i < 8; i++) out[i] = io[i * STARK_HIO_STRIDE]; }
ZKP_HD inline void hio_msg_f128(uint32_t* io, int at, const f128& a) {
    hio_msg(io, at, (uint32_t)a.lo); hio_msg(io, at + 1, (uint32_t)(a.lo >> 32)); hio_msg(io, at + 2, (uint32_t)a.hi); hio_msg(io, at + 3, (uint32_t)(a.hi >> 32));
}
// digest (in the column) of a || b, two 8-word digests
ZKP_HD inline void hio_merge(uint32_t* io, const uint32_t* a, const uint32_t* b) {
    hio_iv(io);
    ZKP_UNROLL for (int i = 0; i < 8; i++) { hio_msg(io, i, a[i]); hio_msg(io, 8 + i, b[i]); }
    blake3_compress_io(io, 64, 11);
}
// digest of seed || u64le(v)
ZKP_HD inline void hio_merge_int(uint32_t* io, const uint32_t* seed, uint64_t v) {
    hio_iv(io);
    ZKP_UNROLL for (int i = 0; i < 8; i++) hio_msg(io, i, seed[i]);
    hio_msg(io, 8, (uint32_t)v); hio_msg(io, 9, (uint32_t)(v >> 32)); hio_msg_zero(io, 10);
    blake3_compress_io(io, 40, 11);
}
// digest of n <= 8 field elements (16 n bytes: one or two blocks of one chunk)
ZKP_HD inline void hio_elements(uint32_t* io, const f128* e, uint32_t n) {
    hio_iv(io);
    const uint32_t nblocks = (n + 3) / 4;
    for (uint32_t b = 0; b < nblocks; b++) {
        const uint32_t have = n - 4 * b < 4 ? n - 4 * b : 4;
        for (uint32_t i = 0; i < 4; i++) { if (i < have) hio_msg_f128(io, 4 * (int)i, e[4 * b + i]); else { hio_msg(io, 4 * (int)i, 0); hio_msg(io, 4 * (int)i + 1, 0); hio_msg(io, 4 * (int)i + 2, 0); hio_msg(io, 4 * (int)i + 3, 0); } }
        blake3_compress_io(io, 16 * have, (b == 0 ? 1u : 0u) | (b == nblocks - 1 ? 10u : 0u));
    }
}

ZKP_HD inline f128 stark_horner8(const f128* c, const f128& x) {
    f128 acc = c[7];
    for (int i = 6; i >= 0; i--) acc = f128_add(f128_mul(acc, x), c[i]);
    return acc;
}
// ---- coin (lane 0; io = lane 0's column)
ZKP_HD inline void coin_reseed(StarkMem& M, uint32_t* io, const uint32_t* d) { hio_merge(io, M.seed, d); hio_digest(M.seed, io); M.counter = 0; }
ZKP_HD inline void coin_reseed_io(StarkMem& M, uint32_t* io) {             // ... with the digest that sits in the column
    uint32_t d[8]; hio_digest(d, io);
    coin_reseed(M, io, d);
}
ZKP_HD inline f128 coin_draw(StarkMem& M, uint32_t* io) {
    for (int tries = 0; tries < 1000; tries++) {
        M.counter++; hio_merge_int(io, M.seed, M.counter);
        f128 v; v.lo = (uint64_t)io[0] | ((uint64_t)io[STARK_HIO_STRIDE] << 32); v.hi = (uint64_t)io[2 * STARK_HIO_STRIDE] | ((uint64_t)io[3 * STARK_HIO_STRIDE] << 32);
        if (!f128_geq_p(v)) return v;
    }
    return f128_make(0);
}

// ---- step 0 (lanes 0..7 useful): trace column; lane 0: coin seed = BLAKE3(context elements || old || new), 160 bytes = three blocks
ZKP_HD inline void stark_step_trace(StarkMem& M, const StarkConst& C, uint64_t oldv, uint64_t newv, uint32_t tid, uint32_t nthreads) {
    const f128 o = f128_make(oldv), n = f128_make(newv);
    const f128 step = f128_mul(f128_sub(n, o), C.inv7);
    for (uint32_t i = tid; i < 8; i += nthreads) {
        f128 v = o; for (uint32_t k = 0; k < i; k++) v = f128_add(v, step);
        M.col[i] = v;
    }
    if (tid == 0) {
        M.step = step; M.oldv = o; M.newv = n;
        uint32_t* io = M.hio;
        hio_iv(io);
        for (int i = 0; i < 16; i++) hio_msg(io, i, C.seed_prefix[i]);
        blake3_compress_io(io, 64, 1);
        for (int i = 0; i < 16; i++) hio_msg(io, i, C.seed_prefix[16 + i]);
        blake3_compress_io(io, 64, 0);
        hio_msg_f128(io, 0, o); hio_msg_f128(io, 4, n); hio_msg_zero(io, 8);
        blake3_compress_io(io, 32, 10);
        hio_digest(M.seed, io); M.counter = 0;
    }
}
ZKP_HD inline void stark_step_interp_trace(StarkMem& M, const StarkConst& C, uint32_t tid, uint32_t nthreads) {
    for (uint32_t k = tid; k < 8; k += nthreads) {
        f128 acc = f128_make(0);
        for (uint32_t i = 0; i < 8; i++) acc = f128_add(acc, f128_mul(M.col[i], C.w8_inv_pow[(i * k) & 7]));
        M.t_poly[k] = f128_mul(acc, C.scale8);
    }
}
// ---- LDE of one column + row hashes (lane = row)
ZKP_HD inline void stark_step_lde(StarkMem& M, const f128* poly, f128* lde, uint32_t (*leaf)[8], const StarkConst& C, uint32_t tid, uint32_t nthreads) {
    for (uint32_t i = tid; i < STARK_LDE; i += nthreads) {
        const f128 v = stark_horner8(poly, C.x_lde[i]);
        lde[i] = v;
        uint32_t* io = M.hio + (i & (STARK_HIO_STRIDE - 1));
        hio_iv(io); hio_msg_f128(io, 0, v); hio_msg_zero(io, 4);
        blake3_compress_io(io, 16, 11);
        hio_digest(leaf[i], io);
    }
}
// one Merkle level: nodes [width, 2*width) from the level below (leaves when width == 32)
ZKP_HD inline void stark_step_merkle_level(StarkMem& M, uint32_t (*leaf)[8], uint32_t (*node)[8], uint32_t width, uint32_t tid, uint32_t nthreads) {
    for (uint32_t j = tid; j < width; j += nthreads) {
        const uint32_t i = width + j;
        uint32_t* io = M.hio + (j & (STARK_HIO_STRIDE - 1));
        if (width == STARK_LDE / 2) hio_merge(io, leaf[2 * j], leaf[2 * j + 1]);
        else hio_merge(io, node[2 * i], node[2 * i + 1]);
        hio_digest(node[i], io);
    }
}
// ---- lane 0: reseed with the trace root, draw the constraint composition coefficients
ZKP_HD inline void stark_step_coefficients(StarkMem& M, uint32_t tid) {
    if (tid != 0) return;
    coin_reseed(M, M.hio, M.t_node[1]);
    for (int i = 0; i < 3; i++) M.coef[i] = coin_draw(M, M.hio);
}
// ---- lanes 0..15: combined constraint evaluations on the constraint-evaluation coset
ZKP_HD inline void stark_step_constraints(StarkMem& M, const StarkConst& C, uint32_t tid, uint32_t nthreads) {
    for (uint32_t i = tid; i < STARK_CE; i += nthreads) {
        const uint32_t row = i * (STARK_LDE / STARK_CE);
        const f128 cur = M.t_lde[row], nxt = M.t_lde[(row + 8) & (STARK_LDE - 1)];
        const f128 t = f128_mul(f128_mul(M.coef[0], f128_sub(f128_sub(nxt, cur), M.step)), C.inv_zt[i]);
        const f128 b0 = f128_mul(f128_mul(M.coef[1], f128_sub(cur, M.oldv)), C.inv_xm1[i]);
        const f128 b1 = f128_mul(f128_mul(M.coef[2], f128_sub(cur, M.newv)), C.inv_xml[i]);
        M.ce[i] = f128_add(f128_add(t, b0), b1);
    }
}
// ---- lanes 0..7: interpolate the 16 evaluations on the coset; the composition column is the low 8 coefficients
ZKP_HD inline void stark_step_interp_constraints(StarkMem& M, const StarkConst& C, uint32_t tid, uint32_t nthreads) {
    for (uint32_t k = tid; k < 8; k += nthreads) {
        f128 acc = f128_make(0);
        for (uint32_t i = 0; i < STARK_CE; i++) acc = f128_add(acc, f128_mul(M.ce[i], C.w16_inv_pow[(i * k) & 15]));
        M.h_poly[k] = f128_mul(acc, C.scale16[k]);
    }
}

// ---- serialisation helpers
struct StarkWriter {
    uint8_t* p; uint32_t n;
    ZKP_HD void u8(uint32_t v) { p[n++] = (uint8_t)v; }
    ZKP_HD void u16(uint32_t v) { u8(v); u8(v >> 8); }
    ZKP_HD void u64(uint64_t v) { for (int i = 0; i < 8; i++) u8((uint32_t)(v >> (8 * i))); }
    ZKP_HD void vint(uint32_t v) { if (v < 128) u8((v << 1) | 1); else if (v < 16384) { const uint32_t e = ((v << 1) | 1) << 1; u8(e); u8(e >> 8); } else { const uint32_t e = ((v << 1) | 1) << 2; u8(e); u8(e >> 8); u8(e >> 16); } }
    ZKP_HD void words(const uint32_t* w, uint32_t k) { for (uint32_t i = 0; i < k; i++) { u8(w[i]); u8(w[i] >> 8); u8(w[i] >> 16); u8(w[i] >> 24); } }
    ZKP_HD void el(const f128& a) { uint32_t w[4]; f128_words(w, a); words(w, 4); }
};
ZKP_HD inline uint32_t stark_vint_len(uint32_t v) { return v < 128 ? 1u : v < 16384 ? 2u : 3u; }
ZKP_HD inline void stark_put_word(uint8_t* p, uint32_t w) { p[0] = (uint8_t)w; p[1] = (uint8_t)(w >> 8); p[2] = (uint8_t)(w >> 16); p[3] = (uint8_t)(w >> 24); }
ZKP_HD inline void stark_put_el(uint8_t* p, const f128& a) { stark_put_word(p, (uint32_t)a.lo); stark_put_word(p + 4, (uint32_t)(a.lo >> 32)); stark_put_word(p + 8, (uint32_t)a.hi); stark_put_word(p + 12, (uint32_t)(a.hi >> 32)); }

// ---- lane 0: the Fiat-Shamir chain after the constraint commitment, down to the seed the query positions are drawn from
ZKP_HD inline void stark_step_chain(StarkMem& M, const StarkConst& C, uint32_t tid) {
    if (tid != 0) return;
    uint32_t* io = M.hio;
    coin_reseed(M, io, M.h_node[1]);
    const f128 z = coin_draw(M, io), zg = f128_mul(z, C.g);
    const f128 tz = stark_horner8(M.t_poly, z), tzg = stark_horner8(M.t_poly, zg);
    M.ood[0] = tz; M.ood[1] = tzg;
    hio_elements(io, M.ood, 2); coin_reseed_io(M, io);
    const f128 hz = stark_horner8(M.h_poly, z);
    M.ood[2] = hz;
    hio_elements(io, M.ood + 2, 1); coin_reseed_io(M, io);
    const f128 dc0 = coin_draw(M, io), dc1 = coin_draw(M, io);
    // DEEP polynomial: dc0 * ((T - T(z))/(x - z) + (T - T(zg))/(x - zg)) + dc1 * (H - H(z))/(x - z), by synthetic division
    {
        f128 c1 = f128_make(0), c2 = f128_make(0), c3 = f128_make(0);
        M.rem[7] = f128_make(0);
        for (int i = 7; i >= 1; i--) {
            c1 = f128_add(M.t_poly[i], f128_mul(c1, z));
            c2 = f128_add(M.t_poly[i], f128_mul(c2, zg));
            c3 = f128_add(M.h_poly[i], f128_mul(c3, z));
            M.rem[i - 1] = f128_add(f128_mul(dc0, f128_add(c1, c2)), f128_mul(dc1, c3));
        }
    }
    hio_elements(io, M.rem, 8); hio_digest(M.rc, io);
    coin_reseed_io(M, io);
    // query positions are drawn from seed' = merge_int(seed, nonce 0)
    hio_merge_int(io, M.seed, 0); hio_digest(M.seed, io); M.counter = 0;
    M.mask = 0;
}
// ---- lanes 0..31: the query positions, 32 independent draws (coin counter q + 1)
ZKP_HD inline void stark_step_queries(StarkMem& M, uint32_t tid, uint32_t nthreads) {
    for (uint32_t q = tid; q < STARK_QUERIES; q += nthreads) {
        uint32_t* io = M.hio + q;
        hio_merge_int(io, M.seed, (uint64_t)q + 1);
        const uint64_t bit = 1ull << (io[0] & (STARK_LDE - 1));
#if defined(__HIP_DEVICE_COMPILE__)
        atomicOr(reinterpret_cast<unsigned long long*>(&M.mask), (unsigned long long)bit);
#else
        M.mask |= bit;
#endif
    }
}
ZKP_HD inline uint32_t stark_popcount64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__popcll(x);
#else
    return (uint32_t)__builtin_popcountll(x);
#endif
}
// ---- lane 0: lays the envelope out -- [2][5][u32 payload][u32 32][old][new][context][count][u16 96][three roots][trace queries]
// [constraint queries][out-of-domain frame][remainder][tail][commitment] -- writes every byte that is not a field element or a digest,
// and lists, once for both openings (same positions, same tree shape), which digests a batch opening carries and where they go.
ZKP_HD inline void stark_step_plan(StarkMem& M, const StarkConst& C, uint64_t oldv, uint64_t newv, uint32_t tid) {
    if (tid != 0) return;
    const uint64_t mask = M.mask;
    const uint32_t count = stark_popcount64(mask);
    M.count = count;
    // which digests: one node list per queried sibling pair (leaf codes 0x80 | index, internal nodes by index)
    uint32_t npairs = 0;
    for (uint32_t pr = 0; pr < STARK_LDE; pr += 2) {
        const uint32_t a = (uint32_t)(mask >> pr) & 1, b = (uint32_t)(mask >> (pr + 1)) & 1;
        if (!(a | b)) continue;
        M.w_cnt[npairs] = 0;
        if (!a) M.w_store[npairs][M.w_cnt[npairs]++] = (uint8_t)(0x80 | pr);
        if (!b) M.w_store[npairs][M.w_cnt[npairs]++] = (uint8_t)(0x80 | (pr + 1));
        M.w_idx[npairs] = (uint8_t)((pr + STARK_LDE) >> 1);
        npairs++;
    }
    uint32_t ncur = npairs;
    for (uint32_t level = 1; level < STARK_DEPTH; level++) {
        uint32_t nn = 0, i = 0;
        while (i < ncur) {
            const uint32_t sib = M.w_idx[i] ^ 1u;
            if (i + 1 < ncur && M.w_idx[i + 1] == sib) i++;
            else M.w_store[i][M.w_cnt[i]++] = (uint8_t)sib;
            M.w_nxt[nn++] = (uint8_t)(sib >> 1);
            i++;
        }
        for (uint32_t k = 0; k < nn; k++) M.w_idx[k] = M.w_nxt[k];
        ncur = nn;
    }
    uint32_t total = 0; for (uint32_t k = 0; k < npairs; k++) total += M.w_cnt[k];
    const uint32_t open_len = 1 + 1 + npairs + 32 * total;            // depth byte, list count (npairs <= 32: one byte), list headers, digests
    const uint32_t block = stark_vint_len(16 * count) + 16 * count + stark_vint_len(open_len) + open_len;
    StarkWriter W{M.out, 0};
    W.u8(2); W.u8(5); W.n += 4; W.u8(32); W.u8(0); W.u8(0); W.u8(0);
    W.u64(oldv); W.u64(newv);
    for (int i = 0; i < 28; i++) W.u8(C.context_bytes[i]);
    W.u8(count);
    W.u16(96); W.n += 96;                                              // the three roots: stark_step_emit
    uint32_t e = 0;
    for (uint32_t tree = 0; tree < 2; tree++) {
        W.vint(16 * count); M.val_off[tree] = W.n; W.n += 16 * count;
        W.vint(open_len); W.u8(STARK_DEPTH); W.vint(npairs);
        M.lists_off[tree] = W.n;
        uint32_t rel = 0;
        for (uint32_t k = 0; k < npairs; k++) {
            W.p[M.lists_off[tree] + rel] = M.w_cnt[k]; rel++;
            for (uint32_t j = 0; j < M.w_cnt[k]; j++) { if (tree == 0) { M.plan_code[e] = M.w_store[k][j]; M.plan_off[e] = (uint16_t)rel; e++; } rel += 32; }
        }
        W.n += rel;
    }
    (void)block;
    M.plan_n = e;
    M.tail_off = W.n;
    W.u16(33); W.u8(2); W.n += 32;                                     // T(z), T(z g)
    W.u16(16); W.n += 16;                                              // H(z)
    W.u8(0); W.u16(128); W.n += 128;                                   // the remainder's eight coefficients
    W.u8(1);
    W.u64(0); W.u8(0);
    const uint32_t payload = W.n - 10;
    M.out[2] = (uint8_t)payload; M.out[3] = (uint8_t)(payload >> 8); M.out[4] = (uint8_t)(payload >> 16); M.out[5] = (uint8_t)(payload >> 24);
    improvement_commitment(M.out + W.n, oldv, newv);
    M.out_len = W.n + 32;
}
// ---- all lanes: field elements and digests into the places the plan gave them
ZKP_HD inline void stark_step_emit(StarkMem& M, uint32_t tid, uint32_t nthreads) {
    const uint64_t mask = M.mask;
    for (uint32_t i = tid; i < STARK_LDE; i += nthreads) {
        if (!((mask >> i) & 1)) continue;
        const uint32_t rank = stark_popcount64(mask & ((1ull << i) - 1));
        stark_put_el(M.out + M.val_off[0] + 16 * rank, M.t_lde[i]);
        stark_put_el(M.out + M.val_off[1] + 16 * rank, M.h_lde[i]);
    }
    const uint32_t nw = M.plan_n * 8;
    for (uint32_t t = tid; t < 2 * nw; t += nthreads) {
        const uint32_t tree = t >= nw ? 1u : 0u, r = t - tree * nw, e = r >> 3, w = r & 7u;
        const uint32_t c = M.plan_code[e];
        const uint32_t v = tree == 0 ? ((c & 0x80) ? M.t_leaf[c & 0x7F][w] : M.t_node[c][w]) : ((c & 0x80) ? M.h_leaf[c & 0x7F][w] : M.h_node[c][w]);
        stark_put_word(M.out + M.lists_off[tree] + M.plan_off[e] + 4 * w, v);
    }
    for (uint32_t t = tid; t < 24; t += nthreads) {                    // the three roots
        const uint32_t v = t < 8 ? M.t_node[1][t] : t < 16 ? M.h_node[1][t - 8] : M.rc[t - 16];
        stark_put_word(M.out + 57 + 4 * t, v);
    }
    for (uint32_t t = tid; t < 11; t += nthreads) {                    // out-of-domain frame and remainder
        const uint32_t at = M.tail_off + (t < 2 ? 3 + 16 * t : t == 2 ? 3 + 32 + 2 : 3 + 32 + 2 + 16 + 3 + 16 * (t - 3));
        stark_put_el(M.out + at, t < 3 ? M.ood[t] : M.rem[t - 3]);
    }
}

// the whole proof; host emulation runs it with nthreads = 1 and a no-op barrier
template <class Sync>
ZKP_HD inline void stark_prove(StarkMem& M, const StarkConst& C, uint64_t oldv, uint64_t newv, uint32_t tid, uint32_t nthreads, Sync sync) {
    stark_step_trace(M, C, oldv, newv, tid, nthreads); sync();
    stark_step_interp_trace(M, C, tid, nthreads); sync();
    stark_step_lde(M, M.t_poly, M.t_lde, M.t_leaf, C, tid, nthreads); sync();
    for (uint32_t w = STARK_LDE / 2; w >= 1; w >>= 1) { stark_step_merkle_level(M, M.t_leaf, M.t_node, w, tid, nthreads); sync(); }
    stark_step_coefficients(M, tid); sync();
    stark_step_constraints(M, C, tid, nthreads); sync();
    stark_step_interp_constraints(M, C, tid, nthreads); sync();
    stark_step_lde(M, M.h_poly, M.h_lde, M.h_leaf, C, tid, nthreads); sync();
    for (uint32_t w = STARK_LDE / 2; w >= 1; w >>= 1) { stark_step_merkle_level(M, M.h_leaf, M.h_node, w, tid, nthreads); sync(); }
    stark_step_chain(M, C, tid); sync();
    stark_step_queries(M, tid, nthreads); sync();
    stark_step_plan(M, C, oldv, newv, tid); sync();
    stark_step_emit(M, tid, nthreads); sync();
}

// ---------------------------------------------------------------------------------------------- the verifier
// proof::improvement_proof::verify_improvement (improvement_proof.rs:37-68) -> StarkBackend::verify (stark.rs:190-211,
// 237-255), restated in oracle/py/stark.py: verify_improvement / verify.  One thread verifies one envelope (a few hundred
// BLAKE3 compressions and field products); every check is a cross-multiplied identity, so no field inversion is needed.
struct StarkReader {
    const uint8_t* b; uint32_t p, n; bool ok;
    ZKP_HD const uint8_t* take(uint32_t k) { if (!ok || n - p < k) { ok = false; return b; } const uint8_t* r = b + p; p += k; return r; }
    ZKP_HD uint32_t u8() { const uint8_t* r = take(1); return ok ? r[0] : 0u; }
    ZKP_HD uint32_t u16() { const uint8_t* r = take(2); return ok ? (uint32_t)r[0] | ((uint32_t)r[1] << 8) : 0u; }
    ZKP_HD uint32_t vint() {                       // winter-utils variable-length usize (values here are < 2^21: at most 3 bytes)
        if (!ok || p >= n) { ok = false; return 0; }
        const uint32_t first = b[p];
        if (first == 0) { ok = false; return 0; }
        uint32_t nb = 1; while (((first >> (nb - 1)) & 1u) == 0) nb++;
        if (nb > 4) { ok = false; return 0; }
        const uint8_t* r = take(nb); if (!ok) return 0;
        uint32_t v = 0; for (uint32_t i = 0; i < nb; i++) v |= (uint32_t)r[i] << (8 * i);
        return v >> nb;
    }
    ZKP_HD f128 element() {
        const uint8_t* r = take(16); f128 v = f128_make(0);
        if (!ok) return v;
        for (int i = 0; i < 8; i++) { v.lo |= (uint64_t)r[i] << (8 * i); v.hi |= (uint64_t)r[8 + i] << (8 * i); }
        if (f128_geq_p(v)) ok = false;
        return v;
    }
};
ZKP_HD inline void ld_digest(uint32_t d[8], const uint8_t* p) { for (int i = 0; i < 8; i++) d[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24); }
struct StarkQueries { f128 val[32]; const uint8_t* list[32]; uint8_t cnt[32]; uint32_t nlists, depth; };
ZKP_HD_NOINLINE inline void stark_read_queries(StarkReader& r, uint32_t nq, StarkQueries& q) {
    if (r.vint() != 16 * nq) r.ok = false;
    for (uint32_t i = 0; i < nq && r.ok; i++) q.val[i] = r.element();
    const uint32_t olen = r.vint(), end = r.p + olen;
    q.depth = r.u8(); q.nlists = r.vint();
    if (q.nlists > 32) r.ok = false;
    for (uint32_t k = 0; k < q.nlists && r.ok; k++) { q.cnt[k] = (uint8_t)r.u8(); q.list[k] = r.take(32u * q.cnt[k]); }
    if (r.p != end) r.ok = false;
}
// root of the batch opening (oracle/py/stark.py: batch_root); false if the node lists do not fit the opened positions
ZKP_HD_NOINLINE inline bool stark_batch_root(uint32_t root[8], const StarkQueries& q, uint64_t mask) {
    uint32_t val[32][8]; uint8_t idx[32], ptr[32];
    uint32_t npairs = 0, k_leaf = 0;
    for (uint32_t pr = 0; pr < STARK_LDE; pr += 2) {
        const uint32_t a = (uint32_t)(mask >> pr) & 1u, b = (uint32_t)(mask >> (pr + 1)) & 1u;
        if (!(a | b)) continue;
        if (npairs >= q.nlists) return false;
        ptr[npairs] = 0;
        uint32_t v[2][8];
        for (uint32_t s = 0; s < 2; s++) {
            if (s == 0 ? a : b) { uint32_t w[4]; f128_words(w, q.val[k_leaf++]); blake3_words(v[s], w, 4); }
            else { if (ptr[npairs] >= q.cnt[npairs]) return false; ld_digest(v[s], q.list[npairs] + 32u * ptr[npairs]); ptr[npairs]++; }
        }
        idx[npairs] = (uint8_t)((pr + STARK_LDE) >> 1);
        blake3_merge(val[npairs], v[0], v[1]);
        npairs++;
    }
    if (npairs != q.nlists) return false;
    uint32_t ncur = npairs;
    for (uint32_t level = 1; level < STARK_DEPTH; level++) {
        uint32_t nn = 0, i = 0;
        while (i < ncur) {
            const uint32_t sib = idx[i] ^ 1u;
            uint32_t out[8];
            if (i + 1 < ncur && idx[i + 1] == sib) { blake3_merge(out, val[i], val[i + 1]); i++; }
            else {
                if (ptr[i] >= q.cnt[i]) return false;
                uint32_t sv[8]; ld_digest(sv, q.list[i] + 32u * ptr[i]); ptr[i]++;
                if ((idx[i] & 1u) == 0) blake3_merge(out, val[i], sv); else blake3_merge(out, sv, val[i]);
            }
            // slot nn <= i - (pairs consumed so far) is never ahead of the read position, so writing in place is safe
            for (int w = 0; w < 8; w++) val[nn][w] = out[w];
            idx[nn] = (uint8_t)(sib >> 1);
            nn++; i++;
        }
        ncur = nn;
    }
    if (ncur != 1) return false;
    for (uint32_t k = 0; k < npairs; k++) if (ptr[k] != q.cnt[k]) return false;
    for (int w = 0; w < 8; w++) root[w] = val[0][w];
    return true;
}
ZKP_HD_NOINLINE inline bool stark_verify_envelope(const uint8_t* env, uint32_t len, uint64_t old_expected, const StarkConst& C) {
    if (len < 10 || env[0] != 2 || env[1] != 5) return false;
    uint32_t plen = 0, clen = 0;
    for (int i = 0; i < 4; i++) { plen |= (uint32_t)env[2 + i] << (8 * i); clen |= (uint32_t)env[6 + i] << (8 * i); }
    if ((uint64_t)10 + plen + clen != len || plen < 16 || clen != 32) return false;
    uint64_t oldv = 0, newv = 0;
    for (int i = 0; i < 8; i++) { oldv |= (uint64_t)env[10 + i] << (8 * i); newv |= (uint64_t)env[18 + i] << (8 * i); }
    if (oldv != old_expected || newv <= oldv) return false;
    uint8_t cm[32]; improvement_commitment(cm, oldv, newv);
    for (int i = 0; i < 32; i++) if (cm[i] != env[10 + plen + i]) return false;
    // ---- parse
    StarkReader r{env + 26, 0, plen - 16, true};
    const uint8_t* ctx = r.take(28);
    if (!r.ok) return false;
    for (int i = 0; i < 28; i++) if (ctx[i] != C.context_bytes[i]) return false;
    const uint32_t nq = r.u8();
    if (r.u16() != 96) return false;
    const uint8_t* roots = r.take(96);
    if (!r.ok || nq < 1 || nq > STARK_QUERIES) return false;
    StarkQueries tq, hq;
    stark_read_queries(r, nq, tq); stark_read_queries(r, nq, hq);
    if (!r.ok || r.u16() != 33 || r.u8() != 2) return false;
    const f128 tz = r.element(), tzg = r.element();
    if (r.u16() != 16) return false;
    const f128 hz = r.element();
    if (r.u8() != 0 || r.u16() != 128) return false;
    f128 rem[8]; for (int i = 0; i < 8; i++) rem[i] = r.element();
    if (r.u8() != 1) return false;
    const uint8_t* tail = r.take(9);
    if (!r.ok || r.p != r.n) return false;
    for (int i = 0; i < 9; i++) if (tail[i] != 0) return false;
    if (tq.depth != STARK_DEPTH || hq.depth != STARK_DEPTH) return false;
    // ---- coin replay
    StarkMem* none = nullptr; (void)none;
    uint32_t seed[8]; uint64_t counter = 0;
    auto reseed = [&](const uint32_t d[8]) { uint32_t s2[8]; blake3_merge(s2, seed, d); for (int i = 0; i < 8; i++) seed[i] = s2[i]; counter = 0; };
    auto draw = [&]() -> f128 {
        for (int tries = 0; tries < 1000; tries++) {
            uint32_t d[8]; counter++; blake3_merge_int(d, seed, counter);
            f128 v; v.lo = (uint64_t)d[0] | ((uint64_t)d[1] << 32); v.hi = (uint64_t)d[2] | ((uint64_t)d[3] << 32);
            if (!f128_geq_p(v)) return v;
        }
        return f128_make(0);
    };
    const f128 o = f128_make(oldv), nw = f128_make(newv), step = f128_mul(f128_sub(nw, o), C.inv7);
    {
        uint32_t w[40];
        for (int i = 0; i < 32; i++) w[i] = C.seed_prefix[i];
        f128_words(w + 32, o); f128_words(w + 36, nw);
        blake3_words(seed, w, 40);
    }
    uint32_t t_root[8], h_root[8], rem_commit[8];
    ld_digest(t_root, roots); ld_digest(h_root, roots + 32); ld_digest(rem_commit, roots + 64);
    reseed(t_root);
    const f128 c0 = draw(), c1 = draw(), c2 = draw();
    reseed(h_root);
    const f128 z = draw(), zg = f128_mul(z, C.g);
    uint32_t d[8];
    { const f128 e[2] = {tz, tzg}; blake3_elements(d, e, 2); reseed(d); }
    // out-of-domain consistency, cross-multiplied by (z^8 - 1)(z - 1)(z - g^7)
    {
        const f128 one = f128_make(1);
        f128 z8 = z; for (int k = 0; k < 3; k++) z8 = f128_mul(z8, z8);
        const f128 a = f128_sub(z8, one), b = f128_sub(z, one), c = f128_sub(z, C.g_last);
        const f128 lhs = f128_mul(f128_mul(hz, a), f128_mul(b, c));
        const f128 t = f128_mul(f128_mul(c0, f128_sub(f128_sub(tzg, tz), step)), f128_mul(f128_mul(c, c), b));
        const f128 b0 = f128_mul(f128_mul(c1, f128_sub(tz, o)), f128_mul(a, c));
        const f128 b1 = f128_mul(f128_mul(c2, f128_sub(tz, nw)), f128_mul(a, b));
        const f128 rhs = f128_add(f128_add(t, b0), b1);
        if (lhs.lo != rhs.lo || lhs.hi != rhs.hi) return false;
    }
    blake3_elements(d, &hz, 1); reseed(d);
    const f128 dc0 = draw(), dc1 = draw();
    blake3_elements(d, rem, 8);
    for (int i = 0; i < 8; i++) if (d[i] != rem_commit[i]) return false;
    reseed(rem_commit);
    { uint32_t s2[8]; blake3_merge_int(s2, seed, 0); for (int i = 0; i < 8; i++) seed[i] = s2[i]; counter = 0; }
    uint64_t mask = 0;
    for (uint32_t q = 0; q < STARK_QUERIES; q++) { counter++; blake3_merge_int(d, seed, counter); mask |= 1ull << (d[0] & (STARK_LDE - 1)); }
    uint32_t count = 0; for (uint32_t i = 0; i < STARK_LDE; i++) count += (uint32_t)(mask >> i) & 1u;
    if (count != nq) return false;
    // ---- Merkle openings
    uint32_t root[8];
    if (!stark_batch_root(root, tq, mask)) return false;
    for (int i = 0; i < 8; i++) if (root[i] != t_root[i]) return false;
    if (!stark_batch_root(root, hq, mask)) return false;
    for (int i = 0; i < 8; i++) if (root[i] != h_root[i]) return false;
    // ---- DEEP composition on the committed remainder polynomial, cross-multiplied by (x - z)(x - z g)
    uint32_t k = 0;
    for (uint32_t pos = 0; pos < STARK_LDE; pos++) {
        if (!((mask >> pos) & 1u)) continue;
        const f128 x = C.x_lde[pos], tv = tq.val[k], hv = hq.val[k];
        k++;
        const f128 xz = f128_sub(x, z), xzg = f128_sub(x, zg);
        const f128 lhs = f128_add(f128_mul(dc0, f128_add(f128_mul(f128_sub(tv, tz), xzg), f128_mul(f128_sub(tv, tzg), xz))), f128_mul(dc1, f128_mul(f128_sub(hv, hz), xzg)));
        const f128 rhs = f128_mul(stark_horner8(rem, x), f128_mul(xz, xzg));
        if (lhs.lo != rhs.lo || lhs.hi != rhs.hi) return false;
    }
    return true;
}

}  // namespace zkp
