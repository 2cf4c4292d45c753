/* ORACLE (test infrastructure only). See transcript.h. */
#include "transcript.h"
#include <string.h>

static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
    0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
    0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
#define ROL(x, n) (((x) << (n)) | ((x) >> (64 - (n))))

void keccak_f1600(uint64_t st[25]) {
    uint64_t bc[5], t;
    for (int r = 0; r < 24; r++) {
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            t = bc[(i + 4) % 5] ^ ROL(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        t = st[1];
        for (int i = 0; i < 24; i++) { int j = PILN[i]; bc[0] = st[j]; st[j] = ROL(t, ROTC[i]); t = bc[0]; }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= RC[r];
    }
}

static void sponge(uint8_t* out, size_t outlen, const uint8_t* in, size_t inlen, size_t rate, uint8_t suffix) {
    union { uint64_t w[25]; uint8_t b[200]; } st;
    memset(&st, 0, sizeof st);
    size_t pos = 0;
    for (size_t i = 0; i < inlen; i++) { st.b[pos++] ^= in[i]; if (pos == rate) { keccak_f1600(st.w); pos = 0; } }
    st.b[pos] ^= suffix; st.b[rate - 1] ^= 0x80;
    keccak_f1600(st.w);
    size_t done = 0;
    while (done < outlen) {
        size_t n = outlen - done < rate ? outlen - done : rate;
        memcpy(out + done, st.b, n); done += n;
        if (done < outlen) keccak_f1600(st.w);
    }
}
void shake256(uint8_t* out, size_t outlen, const uint8_t* in, size_t inlen) { sponge(out, outlen, in, inlen, 136, 0x1F); }
void sha3_512(uint8_t out[64], const uint8_t* in, size_t inlen) { sponge(out, 64, in, inlen, 72, 0x06); }

/* ---------------------------------------------------------------- STROBE-128 / Merlin */
#define STROBE_R 166
enum { FLAG_I = 1, FLAG_A = 2, FLAG_C = 4, FLAG_T = 8, FLAG_M = 16, FLAG_K = 32 };

static void run_f(merlin_t* t) {
    t->st.b[t->pos] ^= t->pos_begin;
    t->st.b[t->pos + 1] ^= 0x04;
    t->st.b[STROBE_R + 1] ^= 0x80;
    keccak_f1600(t->st.w);
    t->pos = 0; t->pos_begin = 0;
}
static void absorb(merlin_t* t, const uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) { t->st.b[t->pos++] ^= d[i]; if (t->pos == STROBE_R) run_f(t); }
}
static void squeeze(merlin_t* t, uint8_t* d, size_t n) {
    for (size_t i = 0; i < n; i++) { d[i] = t->st.b[t->pos]; t->st.b[t->pos++] = 0; if (t->pos == STROBE_R) run_f(t); }
}
static void begin_op(merlin_t* t, uint8_t flags, int more) {
    if (more) return;
    uint8_t hdr[2] = {t->pos_begin, flags};
    t->pos_begin = (uint8_t)(t->pos + 1);
    t->cur_flags = flags;
    absorb(t, hdr, 2);
    if ((flags & (FLAG_C | FLAG_K)) && t->pos != 0) run_f(t);
}
static void meta_ad(merlin_t* t, const uint8_t* d, size_t n, int more) { begin_op(t, FLAG_M | FLAG_A, more); absorb(t, d, n); }
static void ad(merlin_t* t, const uint8_t* d, size_t n, int more) { begin_op(t, FLAG_A, more); absorb(t, d, n); }
static void prf(merlin_t* t, uint8_t* d, size_t n, int more) { begin_op(t, FLAG_I | FLAG_A | FLAG_C, more); squeeze(t, d, n); }

void merlin_init(merlin_t* t, const char* label) {
    memset(t, 0, sizeof *t);
    static const uint8_t hdr[6] = {1, STROBE_R + 2, 1, 0, 1, 96};
    memcpy(t->st.b, hdr, 6);
    memcpy(t->st.b + 6, "STROBEv1.0.2", 12);
    keccak_f1600(t->st.w);
    meta_ad(t, (const uint8_t*)"Merlin v1.0", 11, 0);
    merlin_append(t, "dom-sep", (const uint8_t*)label, (uint32_t)strlen(label));
}
void merlin_append(merlin_t* t, const char* label, const uint8_t* msg, uint32_t len) {
    uint8_t le[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    meta_ad(t, (const uint8_t*)label, strlen(label), 0);
    meta_ad(t, le, 4, 1);
    ad(t, msg, len, 0);
}
void merlin_append_u64(merlin_t* t, const char* label, uint64_t x) {
    uint8_t b[8]; for (int i = 0; i < 8; i++) b[i] = (uint8_t)(x >> (8 * i));
    merlin_append(t, label, b, 8);
}
void merlin_challenge(merlin_t* t, const char* label, uint8_t* out, uint32_t len) {
    uint8_t le[4] = {(uint8_t)len, (uint8_t)(len >> 8), (uint8_t)(len >> 16), (uint8_t)(len >> 24)};
    meta_ad(t, (const uint8_t*)label, strlen(label), 0);
    meta_ad(t, le, 4, 1);
    prf(t, out, len, 0);
}

/* ---------------------------------------------------------------- SHA-256 (FIPS 180-4), for prove_consistency's digest
 * (/root/reference/src/backend/bulletproofs.rs:434) */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR32(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(uint32_t h[8], const uint8_t* p) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR32(w[i - 15], 7) ^ ROR32(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ROR32(w[i - 2], 17) ^ ROR32(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = ROR32(e, 6) ^ ROR32(e, 11) ^ ROR32(e, 25), ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + K256[i] + w[i];
        uint32_t S0 = ROR32(a, 2) ^ ROR32(a, 13) ^ ROR32(a, 22), mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
void sha256(uint8_t out[32], const uint8_t* in, size_t inlen) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t i = 0;
    for (; i + 64 <= inlen; i += 64) sha256_block(h, in + i);
    uint8_t buf[128]; size_t rem = inlen - i;
    memset(buf, 0, sizeof buf);
    memcpy(buf, in + i, rem);
    buf[rem] = 0x80;
    size_t tot = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)inlen * 8;
    for (int k = 0; k < 8; k++) buf[tot - 1 - k] = (uint8_t)(bits >> (8 * k));
    sha256_block(h, buf);
    if (tot == 128) sha256_block(h, buf + 64);
    for (int k = 0; k < 8; k++) { out[4 * k] = (uint8_t)(h[k] >> 24); out[4 * k + 1] = (uint8_t)(h[k] >> 16); out[4 * k + 2] = (uint8_t)(h[k] >> 8); out[4 * k + 3] = (uint8_t)h[k]; }
}
