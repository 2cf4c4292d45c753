/* ORACLE (test infrastructure only; never linked into or called by the product path).
 *
 * CPU restatement in plain C of libzkp's improvement proof: the STARK the reference obtains from winterfell ^0.10
 * (Cargo.toml:22-23; not vendored, not pinned, not buildable here) through /root/reference/src/backend/stark.rs:15-186, with
 * libzkp's framing (improvement_proof.rs:10-35, utils/commitment.rs:38-50).  PARITY UNPINNED: see the header of
 * oracle/py/stark.py, which lists every detail of Winterfell's coin / Merkle / serialisation that is restated from its
 * published design without a source or vector to check against.  This file is a line-by-line C port of that Python model
 * and is pinned to it bit for bit (tests/test_oracle_c_snark.py, tests/golden/stark_oracle_vectors.json).
 */
#include "zkp_oracle.h"
#include "transcript.h"
#include <string.h>
#include <stdlib.h>

typedef unsigned __int128 u128;

/* ---------------------------------------------------------------- f128: p = 2^128 - 45 * 2^40 + 1 (stark.rs:6) */
#define P128 ((((u128)0xFFFFFFFFFFFFFFFFull) << 64 | 0xFFFFFFFFFFFFFFFFull) - ((u128)45 << 40) + 2)   /* 2^128 - 45*2^40 + 1 */
static const u128 FOLD = ((u128)45 << 40) - 1;                                     /* 2^128 mod p */
static u128 f_add(u128 a, u128 b) { u128 s = a + b; if (s < a) s += FOLD; if (s >= P128) s -= P128; return s; }
static u128 f_sub(u128 a, u128 b) { return a >= b ? a - b : a + (P128 - b); }
static void mul_wide(u128 a, u128 b, u128* hi, u128* lo) {
    const uint64_t a0 = (uint64_t)a, a1 = (uint64_t)(a >> 64), b0 = (uint64_t)b, b1 = (uint64_t)(b >> 64);
    const u128 p00 = (u128)a0 * b0, p01 = (u128)a0 * b1, p10 = (u128)a1 * b0, p11 = (u128)a1 * b1;
    const u128 mid = (p00 >> 64) + (uint64_t)p01 + (uint64_t)p10;
    *lo = (mid << 64) | (uint64_t)p00;
    *hi = p11 + (p01 >> 64) + (p10 >> 64) + (mid >> 64);
}
static u128 f_mul(u128 a, u128 b) {
    u128 hi, lo, h2, l2;
    mul_wide(a, b, &hi, &lo);
    mul_wide(hi, FOLD, &h2, &l2);                  /* hi * 2^128 = hi * FOLD (mod p); h2 < 2^46 */
    u128 r = lo + l2; u128 carry = r < lo;
    u128 t = h2 * FOLD;                            /* < 2^92 */
    u128 r2 = r + t; carry += r2 < r;
    while (carry) { u128 r3 = r2 + FOLD; carry = carry - 1 + (r3 < r2); r2 = r3; }
    if (r2 >= P128) r2 -= P128;
    return r2;
}
static u128 f_pow(u128 a, u128 e) { u128 acc = 1; while (e) { if (e & 1) acc = f_mul(acc, a); a = f_mul(a, a); e >>= 1; } return acc; }
static u128 f_inv(u128 a) { return f_pow(a, P128 - 2); }
static u128 root_of_unity(uint32_t n) {            /* 3^((p-1)/2^40) raised to 2^40 / n */
    const u128 two_adic = f_pow(3, (P128 - 1) >> 40);
    return f_pow(two_adic, ((u128)1 << 40) / n);
}
static void el_bytes(uint8_t b[16], u128 x) { for (int i = 0; i < 16; i++) b[i] = (uint8_t)(x >> (8 * i)); }

/* ---------------------------------------------------------------- BLAKE3, single chunk (<= 1024 bytes), 32-byte output */
static const uint32_t B3_IV[8] = {0x6A09E667, 0xBB67AE85, 0x3C6EF372, 0xA54FF53A, 0x510E527F, 0x9B05688C, 0x1F83D9AB, 0x5BE0CD19};
static const uint8_t B3_PERM[16] = {2, 6, 3, 10, 7, 0, 4, 13, 1, 11, 12, 5, 9, 14, 15, 8};
static uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void b3_g(uint32_t* s, int a, int b, int c, int d, uint32_t mx, uint32_t my) {
    s[a] = s[a] + s[b] + mx; s[d] = rotr32(s[d] ^ s[a], 16);
    s[c] = s[c] + s[d]; s[b] = rotr32(s[b] ^ s[c], 12);
    s[a] = s[a] + s[b] + my; s[d] = rotr32(s[d] ^ s[a], 8);
    s[c] = s[c] + s[d]; s[b] = rotr32(s[b] ^ s[c], 7);
}
static void b3_compress(uint32_t cv[8], const uint32_t block[16], uint32_t block_len, uint32_t flags) {
    uint32_t s[16], m[16], t[16];
    memcpy(s, cv, 32); memcpy(s + 8, B3_IV, 16); s[12] = 0; s[13] = 0; s[14] = block_len; s[15] = flags;
    memcpy(m, block, 64);
    for (int r = 0; r < 7; r++) {
        b3_g(s, 0, 4, 8, 12, m[0], m[1]); b3_g(s, 1, 5, 9, 13, m[2], m[3]); b3_g(s, 2, 6, 10, 14, m[4], m[5]); b3_g(s, 3, 7, 11, 15, m[6], m[7]);
        b3_g(s, 0, 5, 10, 15, m[8], m[9]); b3_g(s, 1, 6, 11, 12, m[10], m[11]); b3_g(s, 2, 7, 8, 13, m[12], m[13]); b3_g(s, 3, 4, 9, 14, m[14], m[15]);
        if (r < 6) { for (int i = 0; i < 16; i++) t[i] = m[B3_PERM[i]]; memcpy(m, t, 64); }
    }
    for (int i = 0; i < 8; i++) cv[i] = s[i] ^ s[i + 8];
}
static void blake3(uint8_t out[32], const uint8_t* data, size_t len) {
    uint32_t cv[8]; memcpy(cv, B3_IV, 32);
    const size_t nblocks = len ? (len + 63) / 64 : 1;
    for (size_t i = 0; i < nblocks; i++) {
        uint8_t blk[64]; memset(blk, 0, 64);
        const size_t bl = i + 1 < nblocks ? 64 : len - 64 * i;
        memcpy(blk, data + 64 * i, bl);
        uint32_t w[16];
        for (int k = 0; k < 16; k++) w[k] = (uint32_t)blk[4 * k] | ((uint32_t)blk[4 * k + 1] << 8) | ((uint32_t)blk[4 * k + 2] << 16) | ((uint32_t)blk[4 * k + 3] << 24);
        const uint32_t flags = (i == 0 ? 1u : 0u) | (i + 1 == nblocks ? (2u | 8u) : 0u);      /* CHUNK_START | CHUNK_END | ROOT */
        b3_compress(cv, w, (uint32_t)bl, flags);
    }
    for (int k = 0; k < 8; k++) { out[4 * k] = (uint8_t)cv[k]; out[4 * k + 1] = (uint8_t)(cv[k] >> 8); out[4 * k + 2] = (uint8_t)(cv[k] >> 16); out[4 * k + 3] = (uint8_t)(cv[k] >> 24); }
}
static void hash_elements(uint8_t out[32], const u128* e, uint32_t n) {
    uint8_t buf[16 * 16];
    for (uint32_t i = 0; i < n; i++) el_bytes(buf + 16 * i, e[i]);
    blake3(out, buf, 16 * n);
}
static void merge(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]) { uint8_t buf[64]; memcpy(buf, a, 32); memcpy(buf + 32, b, 32); blake3(out, buf, 64); }
static void merge_with_int(uint8_t out[32], const uint8_t seed[32], uint64_t v) {
    uint8_t buf[40]; memcpy(buf, seed, 32); for (int i = 0; i < 8; i++) buf[32 + i] = (uint8_t)(v >> (8 * i));
    blake3(out, buf, 40);
}

/* ---------------------------------------------------------------- random coin (oracle/py/stark.py: Coin) */
typedef struct { uint8_t seed[32]; uint64_t counter; } coin_t;
static void coin_reseed(coin_t* c, const uint8_t d[32]) { uint8_t s[32]; merge(s, c->seed, d); memcpy(c->seed, s, 32); c->counter = 0; }
static u128 coin_draw(coin_t* c) {
    for (;;) {
        uint8_t h[32]; c->counter++; merge_with_int(h, c->seed, c->counter);
        u128 v = 0; for (int i = 15; i >= 0; i--) v = (v << 8) | h[i];
        if (v < P128) return v;
    }
}

#define TRACE_LEN 8
#define BLOWUP 8
#define LDE 64
#define CE 16
#define NQ 32
#define DEPTH 6

static u128 horner(const u128* c, uint32_t n, u128 x) { u128 acc = 0; for (int i = (int)n - 1; i >= 0; i--) acc = f_add(f_mul(acc, x), c[i]); return acc; }
/* values on offset * <w_n> (natural order) -> n coefficients */
static void interpolate(u128* out, const u128* ev, uint32_t n, u128 offset) {
    const u128 w_inv = f_inv(root_of_unity(n)), n_inv = f_inv(n), o_inv = f_inv(offset);
    u128 wp[64]; wp[0] = 1; for (uint32_t i = 1; i < n; i++) wp[i] = f_mul(wp[i - 1], w_inv);
    u128 ok = 1;
    for (uint32_t k = 0; k < n; k++) {
        u128 acc = 0;
        for (uint32_t i = 0; i < n; i++) acc = f_add(acc, f_mul(ev[i], wp[(i * k) % n]));
        out[k] = f_mul(f_mul(acc, n_inv), ok);
        ok = f_mul(ok, o_inv);
    }
}
static void evaluate_on_coset(u128* out, const u128* coefs, uint32_t nc, uint32_t n, u128 offset) {
    const u128 w = root_of_unity(n); u128 x = offset;
    for (uint32_t i = 0; i < n; i++) { out[i] = horner(coefs, nc, x); x = f_mul(x, w); }
}
/* poly(x) / (x - a), remainder dropped: out has n - 1 coefficients */
static void syn_div(u128* out, const u128* coefs, uint32_t n, u128 a) {
    u128 carry = 0;
    for (int i = (int)n - 1; i >= 1; i--) { carry = f_add(coefs[i], f_mul(carry, a)); out[i - 1] = carry; }
}

/* Merkle tree: node i has children 2i, 2i + 1; leaves hang below nodes n/2 .. n-1 */
typedef struct { uint8_t leaves[LDE][32]; uint8_t nodes[LDE][32]; } tree_t;
static void tree_build(tree_t* T) {
    for (int i = LDE - 1; i >= 1; i--) {
        if (i >= LDE / 2) merge(T->nodes[i], T->leaves[2 * (i - LDE / 2)], T->leaves[2 * (i - LDE / 2) + 1]);
        else merge(T->nodes[i], T->nodes[2 * i], T->nodes[2 * i + 1]);
    }
}
/* batch opening of sorted unique positions: per opened pair one node list (MerkleTree.prove_batch); serialised on the fly */
typedef struct { uint8_t n; uint8_t d[DEPTH][32]; } nodelist;
static uint32_t tree_prove_batch(const tree_t* T, const uint32_t* pos, uint32_t np, nodelist* lists) {
    uint8_t opened[LDE]; memset(opened, 0, sizeof opened);
    for (uint32_t k = 0; k < np; k++) opened[pos[k]] = 1;
    uint32_t cur[NQ], nxt[NQ], npairs = 0;
    for (uint32_t k = 0; k < np; k++) {
        const uint32_t p = pos[k] & ~1u;
        if (npairs && cur[npairs - 1] == p) continue;              /* positions are sorted: equal pairs are adjacent */
        cur[npairs++] = p;
    }
    for (uint32_t k = 0; k < npairs; k++) {
        lists[k].n = 0;
        for (uint32_t i = cur[k]; i < cur[k] + 2; i++) if (!opened[i]) memcpy(lists[k].d[lists[k].n++], T->leaves[i], 32);
        nxt[k] = (cur[k] + LDE) >> 1;
    }
    uint32_t ncur = npairs;
    for (int level = 1; level < DEPTH; level++) {
        memcpy(cur, nxt, sizeof(uint32_t) * ncur);
        uint32_t nn = 0, i = 0;
        while (i < ncur) {
            const uint32_t sib = cur[i] ^ 1u;
            if (i + 1 < ncur && cur[i + 1] == sib) i++;
            else memcpy(lists[i].d[lists[i].n++], T->nodes[sib], 32);
            nxt[nn++] = sib >> 1;
            i++;
        }
        ncur = nn;
    }
    return npairs;
}
static uint32_t put_vint(uint8_t* o, uint64_t n) {                 /* winter-utils variable-length usize */
    uint32_t bits = 0; for (uint64_t t = n; t; t >>= 1) bits++; if (bits == 0) bits = 1;
    const uint32_t nbytes = (bits + 6) / 7;
    if (nbytes > 8) { o[0] = 0; for (int i = 0; i < 8; i++) o[1 + i] = (uint8_t)(n >> (8 * i)); return 9; }
    const u128 v = (((u128)n << 1) | 1) << (nbytes - 1);
    for (uint32_t i = 0; i < nbytes; i++) o[i] = (uint8_t)(v >> (8 * i));
    return nbytes;
}
static uint32_t put_queries(uint8_t* o, const u128* values, uint32_t np, const nodelist* lists, uint32_t nlists) {
    uint32_t p = put_vint(o, 16ull * np);
    for (uint32_t k = 0; k < np; k++) { el_bytes(o + p, values[k]); p += 16; }
    uint8_t ob[2 + 9 + NQ * (1 + DEPTH * 32)]; uint32_t q = 0;
    ob[q++] = DEPTH; q += put_vint(ob + q, nlists);
    for (uint32_t k = 0; k < nlists; k++) { ob[q++] = lists[k].n; memcpy(ob + q, lists[k].d, 32u * lists[k].n); q += 32u * lists[k].n; }
    p += put_vint(o + p, q); memcpy(o + p, ob, q); p += q;
    return p;
}

/* StarkBackend::prove_improvement (stark.rs:151-186): the bare proof bytes; returns the length */
static uint32_t stark_prove(uint64_t old, uint64_t new_, uint8_t* out) {
    const u128 g = root_of_unity(TRACE_LEN);
    /* trace old, old + s, ..., new with s = (new - old) / 7 in the field (stark.rs:161-165) */
    const u128 step = f_mul(f_sub(new_, old), f_inv(TRACE_LEN - 1));
    u128 col[TRACE_LEN]; for (int i = 0; i < TRACE_LEN; i++) col[i] = f_add(old, f_mul((u128)i, step));
    const u128 modulus = P128;
    u128 seed_el[10] = {(1u << 8) | 0, TRACE_LEN, (uint64_t)modulus, (uint64_t)(modulus >> 64),
                        (1u << 16) | (8u << 8) | 31u, 0, BLOWUP, NQ, old, new_};
    coin_t coin; hash_elements(coin.seed, seed_el, 10); coin.counter = 0;
    /* 1. trace commitment */
    u128 t_poly[TRACE_LEN], t_lde[LDE];
    interpolate(t_poly, col, TRACE_LEN, 1);
    evaluate_on_coset(t_lde, t_poly, TRACE_LEN, LDE, 3);
    tree_t* tt = (tree_t*)malloc(sizeof(tree_t)); tree_t* ht = (tree_t*)malloc(sizeof(tree_t));
    for (int i = 0; i < LDE; i++) hash_elements(tt->leaves[i], &t_lde[i], 1);
    tree_build(tt);
    coin_reseed(&coin, tt->nodes[1]);
    /* 2. constraint evaluations on the 16-point coset, divided by their divisors */
    u128 coef[3]; for (int i = 0; i < 3; i++) coef[i] = coin_draw(&coin);
    const u128 w_ce = root_of_unity(CE), last = f_pow(g, TRACE_LEN - 1);
    u128 ce[CE], x = 3;
    for (int i = 0; i < CE; i++) {
        const u128 cur = t_lde[i * (LDE / CE)], nxt = t_lde[(i * (LDE / CE) + BLOWUP) % LDE];
        const u128 z_t = f_mul(f_sub(f_pow(x, TRACE_LEN), 1), f_inv(f_sub(x, last)));
        const u128 t = f_mul(f_mul(coef[0], f_sub(f_sub(nxt, cur), step)), f_inv(z_t));
        const u128 b0 = f_mul(f_mul(coef[1], f_sub(cur, old)), f_inv(f_sub(x, 1)));
        const u128 b1 = f_mul(f_mul(coef[2], f_sub(cur, new_)), f_inv(f_sub(x, last)));
        ce[i] = f_add(f_add(t, b0), b1);
        x = f_mul(x, w_ce);
    }
    u128 h_full[CE], h_lde[LDE];
    interpolate(h_full, ce, CE, 3);
    const u128* h_poly = h_full;                                  /* degrees 8..15 vanish: one composition column */
    evaluate_on_coset(h_lde, h_poly, TRACE_LEN, LDE, 3);
    for (int i = 0; i < LDE; i++) hash_elements(ht->leaves[i], &h_lde[i], 1);
    tree_build(ht);
    coin_reseed(&coin, ht->nodes[1]);
    /* 3. out-of-domain frame */
    const u128 z = coin_draw(&coin), zg = f_mul(z, g);
    const u128 tz = horner(t_poly, TRACE_LEN, z), tzg = horner(t_poly, TRACE_LEN, zg);
    uint8_t d[32];
    { u128 e2[2] = {tz, tzg}; hash_elements(d, e2, 2); coin_reseed(&coin, d); }
    const u128 hz = horner(h_poly, TRACE_LEN, z);
    hash_elements(d, &hz, 1); coin_reseed(&coin, d);
    /* 4. DEEP composition polynomial */
    u128 deep[2] = {coin_draw(&coin), 0}; deep[1] = coin_draw(&coin);
    u128 num[TRACE_LEN], t1[TRACE_LEN - 1], t2[TRACE_LEN - 1], c1[TRACE_LEN - 1], rem[TRACE_LEN];
    memcpy(num, t_poly, sizeof num); num[0] = f_sub(t_poly[0], tz); syn_div(t1, num, TRACE_LEN, z);
    num[0] = f_sub(t_poly[0], tzg); syn_div(t2, num, TRACE_LEN, zg);
    memcpy(num, h_poly, sizeof num); num[0] = f_sub(h_poly[0], hz); syn_div(c1, num, TRACE_LEN, z);
    for (int i = 0; i < TRACE_LEN - 1; i++) rem[i] = f_add(f_mul(deep[0], f_add(t1[i], t2[i])), f_mul(deep[1], c1[i]));
    rem[TRACE_LEN - 1] = 0;
    /* 5. FRI with zero folding layers: the remainder is the polynomial itself, committed by its hash */
    uint8_t rem_commit[32]; hash_elements(rem_commit, rem, TRACE_LEN);
    coin_reseed(&coin, rem_commit);
    /* 6. query positions: nonce 0, 32 draws, sorted and deduplicated */
    { uint8_t s[32]; merge_with_int(s, coin.seed, 0); memcpy(coin.seed, s, 32); coin.counter = 0; }
    uint64_t mask = 0;
    for (int k = 0; k < NQ; k++) {
        uint8_t h[32]; coin.counter++; merge_with_int(h, coin.seed, coin.counter);
        uint64_t v = 0; for (int i = 7; i >= 0; i--) v = (v << 8) | h[i];
        mask |= 1ull << (v & (LDE - 1));
    }
    uint32_t pos[NQ], np = 0; for (uint32_t i = 0; i < LDE; i++) if ((mask >> i) & 1) pos[np++] = i;
    /* 7. serialisation (oracle/py/stark.py: context_bytes, _queries_bytes, prove) */
    uint32_t p = 0;
    out[p++] = 1; out[p++] = 0; out[p++] = 3; out[p++] = 0; out[p++] = 0; out[p++] = 16;
    el_bytes(out + p, modulus); p += 16;
    out[p++] = NQ; out[p++] = BLOWUP; out[p++] = 0; out[p++] = 1; out[p++] = 8; out[p++] = 31;
    out[p++] = (uint8_t)np;
    out[p++] = 96; out[p++] = 0;
    memcpy(out + p, tt->nodes[1], 32); p += 32; memcpy(out + p, ht->nodes[1], 32); p += 32; memcpy(out + p, rem_commit, 32); p += 32;
    nodelist lists[NQ]; u128 vals[NQ];
    uint32_t nl = tree_prove_batch(tt, pos, np, lists);
    for (uint32_t k = 0; k < np; k++) vals[k] = t_lde[pos[k]];
    p += put_queries(out + p, vals, np, lists, nl);
    nl = tree_prove_batch(ht, pos, np, lists);
    for (uint32_t k = 0; k < np; k++) vals[k] = h_lde[pos[k]];
    p += put_queries(out + p, vals, np, lists, nl);
    out[p++] = 33; out[p++] = 0; out[p++] = 2; el_bytes(out + p, tz); p += 16; el_bytes(out + p, tzg); p += 16;
    out[p++] = 16; out[p++] = 0; el_bytes(out + p, hz); p += 16;
    out[p++] = 0; out[p++] = 128; out[p++] = 0;
    for (int i = 0; i < TRACE_LEN; i++) { el_bytes(out + p, rem[i]); p += 16; }
    out[p++] = 1;
    memset(out + p, 0, 8); p += 8;
    out[p++] = 0;
    free(tt); free(ht);
    return p;
}

/* proof::improvement_proof::prove_improvement (improvement_proof.rs:10-35): [2][5][u32 len][u32 32][old || new || stark][SHA-256 binding] */
int zkp_oracle_prove_improvement(uint64_t old, uint64_t new_, uint8_t* out, uint32_t cap, uint32_t* out_len) {
    *out_len = 0;
    if (new_ <= old) return ZKP_ORACLE_INVALID_INPUT;             /* validation.rs:63-71 */
    if (cap < 3527) return ZKP_ORACLE_BUFFER_TOO_SMALL;
    uint8_t* pay = out + 10;
    for (int i = 0; i < 8; i++) { pay[i] = (uint8_t)(old >> (8 * i)); pay[8 + i] = (uint8_t)(new_ >> (8 * i)); }
    const uint32_t sl = stark_prove(old, new_, pay + 16), plen = 16 + sl;
    out[0] = 2; out[1] = 5;
    for (int i = 0; i < 4; i++) { out[2 + i] = (uint8_t)(plen >> (8 * i)); out[6 + i] = (uint8_t)(32u >> (8 * i)); }
    uint8_t in[21 + 16]; memcpy(in, "libzkp_improvement_v1", 21); memcpy(in + 21, pay, 16);      /* utils/commitment.rs:38-50 */
    sha256(out + 10 + plen, in, sizeof in);
    *out_len = 10 + plen + 32;
    return 0;
}
