/* ORACLE (test infrastructure only). See bn254.h for scope and the "parity unpinned" statement.
 * BN254 Fq / Fr in 4 x 64-bit Montgomery limbs (CIOS), Fq2 = Fq[u]/(u^2 + 1), G1 / G2 in Jacobian coordinates, the
 * bucket-method MSM that ark-ec's VariableBaseMSM uses, ark-serialize's uncompressed point encoding. */
#include "bn254.h"
#include <string.h>
#include <stdlib.h>

typedef unsigned __int128 u128;

typedef struct { uint64_t p[4]; uint64_t inv; fp r2, one; } modctx;
static modctx MQ, MR;
fp FR_ONE, FR_ZERO;
static int g_ready = 0;

static const uint64_t P_Q[4] = {0x3c208c16d87cfd47ull, 0x97816a916871ca8dull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
static const uint64_t P_R[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};

static int geq(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--) { if (a[i] > b[i]) return 1; if (a[i] < b[i]) return 0; }
    return 1;
}
static uint64_t sub4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) { u128 d = (u128)a[i] - b[i] - br; r[i] = (uint64_t)d; br = (uint64_t)(d >> 64) & 1; }
    return br;
}
static uint64_t add4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t c = 0;
    for (int i = 0; i < 4; i++) { u128 s = (u128)a[i] + b[i] + c; r[i] = (uint64_t)s; c = (uint64_t)(s >> 64); }
    return c;
}
static void m_add(const modctx* M, fp* r, const fp* a, const fp* b) {
    uint64_t t[4]; const uint64_t c = add4(t, a->v, b->v);
    if (c || geq(t, M->p)) sub4(t, t, M->p);
    memcpy(r->v, t, 32);
}
static void m_sub(const modctx* M, fp* r, const fp* a, const fp* b) {
    uint64_t t[4];
    if (sub4(t, a->v, b->v)) add4(t, t, M->p);
    memcpy(r->v, t, 32);
}
/* Montgomery product, CIOS without the extra carry word (valid because the top bit of both BN254 moduli is clear); the
 * modulus and -p^-1 mod 2^64 are compile-time constants at every call site so the loops unroll into straight-line code */
#define INV_Q 0x87d20782e4866389ull
#define INV_R 0xc2e1f593efffffffull
static inline __attribute__((always_inline)) void mont_mul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4], const uint64_t p[4], const uint64_t inv) {
    uint64_t t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    for (int i = 0; i < 4; i++) {
        u128 A = (u128)a[0] * b[i] + t0;
        const uint64_t m = (uint64_t)A * inv;
        u128 C = (u128)m * p[0] + (uint64_t)A;
        A = (u128)a[1] * b[i] + t1 + (uint64_t)(A >> 64); C = (u128)m * p[1] + (uint64_t)A + (uint64_t)(C >> 64); t0 = (uint64_t)C;
        A = (u128)a[2] * b[i] + t2 + (uint64_t)(A >> 64); C = (u128)m * p[2] + (uint64_t)A + (uint64_t)(C >> 64); t1 = (uint64_t)C;
        A = (u128)a[3] * b[i] + t3 + (uint64_t)(A >> 64); C = (u128)m * p[3] + (uint64_t)A + (uint64_t)(C >> 64); t2 = (uint64_t)C;
        t3 = (uint64_t)(C >> 64) + (uint64_t)(A >> 64);
    }
    uint64_t t[4] = {t0, t1, t2, t3}, u[4];
    const uint64_t borrow = sub4(u, t, p);
    if (!borrow) memcpy(r, u, 32); else memcpy(r, t, 32);
}
static void mq_mul(fp* r, const fp* a, const fp* b) { mont_mul(r->v, a->v, b->v, P_Q, INV_Q); }
static void mr_mul(fp* r, const fp* a, const fp* b) { mont_mul(r->v, a->v, b->v, P_R, INV_R); }
static void m_mul(const modctx* M, fp* r, const fp* a, const fp* b) { if (M == &MQ) mq_mul(r, a, b); else mr_mul(r, a, b); }
static void m_pow(const modctx* M, fp* r, const fp* a, const uint64_t e[4]) {
    fp acc = M->one, base = *a;
    for (int i = 0; i < 256; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) m_mul(M, &acc, &acc, &base);
        m_mul(M, &base, &base, &base);
    }
    *r = acc;
}
static void m_inv(const modctx* M, fp* r, const fp* a) {
    uint64_t e[4]; const uint64_t two[4] = {2, 0, 0, 0};
    sub4(e, M->p, two);
    m_pow(M, r, a, e);
}
static void m_from_raw(const modctx* M, fp* r, const uint64_t w[4]) { fp t; memcpy(t.v, w, 32); m_mul(M, r, &t, &M->r2); }   /* valid for any w < 2^256 */
static void m_to_raw(const modctx* M, uint64_t w[4], const fp* a) { fp one = {{1, 0, 0, 0}}, t; m_mul(M, &t, a, &one); memcpy(w, t.v, 32); }
static void ctx_init(modctx* M, const uint64_t p[4]) {
    memcpy(M->p, p, 32);
    uint64_t x = 1;                                           /* -p^-1 mod 2^64 by Newton iteration */
    for (int i = 0; i < 6; i++) x *= 2 - p[0] * x;
    M->inv = (uint64_t)0 - x;
    if (M->inv != (p == P_Q ? INV_Q : INV_R)) abort();       /* the hard-coded constants above are -p^-1 mod 2^64 */
    fp t = {{1, 0, 0, 0}};                                    /* 2^512 mod p by 512 modular doublings */
    fp one_m;
    for (int i = 0; i < 512; i++) {
        m_add(M, &t, &t, &t);
        if (i == 255) one_m = t;
    }
    M->r2 = t; M->one = one_m;
}

static void le_to_limbs(uint64_t w[4], const uint8_t b[32]) { for (int i = 0; i < 4; i++) { uint64_t x = 0; for (int k = 7; k >= 0; k--) x = (x << 8) | b[8 * i + k]; w[i] = x; } }
static void limbs_to_le(uint8_t b[32], const uint64_t w[4]) { for (int i = 0; i < 4; i++) for (int k = 0; k < 8; k++) b[8 * i + k] = (uint8_t)(w[i] >> (8 * k)); }

void bn254_init(void) {
    if (g_ready) return;
    ctx_init(&MQ, P_Q); ctx_init(&MR, P_R);
    FR_ONE = MR.one; memset(&FR_ZERO, 0, sizeof FR_ZERO);
    g_ready = 1;
}

/* ---------------------------------------------------------------- Fq */
void fq_add(fp* r, const fp* a, const fp* b) { m_add(&MQ, r, a, b); }
void fq_sub(fp* r, const fp* a, const fp* b) { m_sub(&MQ, r, a, b); }
void fq_neg(fp* r, const fp* a) { fp z; memset(&z, 0, sizeof z); m_sub(&MQ, r, &z, a); }
void fq_mul(fp* r, const fp* a, const fp* b) { mq_mul(r, a, b); }
void fq_inv(fp* r, const fp* a) { m_inv(&MQ, r, a); }
int fq_is_zero(const fp* a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
int fq_from_bytes(fp* r, const uint8_t b[32]) {
    uint64_t w[4]; le_to_limbs(w, b);
    if (geq(w, MQ.p)) return 0;
    m_from_raw(&MQ, r, w);
    return 1;
}
void fq_to_bytes(uint8_t b[32], const fp* a) { uint64_t w[4]; m_to_raw(&MQ, w, a); limbs_to_le(b, w); }
int fq_lex_larger(const fp* a) {
    uint64_t w[4], n[4]; m_to_raw(&MQ, w, a);
    if ((w[0] | w[1] | w[2] | w[3]) == 0) return 0;
    sub4(n, MQ.p, w);
    return !geq(n, w);                                       /* w > p - w */
}

/* ---------------------------------------------------------------- Fr */
void fr_add(fp* r, const fp* a, const fp* b) { m_add(&MR, r, a, b); }
void fr_sub(fp* r, const fp* a, const fp* b) { m_sub(&MR, r, a, b); }
void fr_neg(fp* r, const fp* a) { m_sub(&MR, r, &FR_ZERO, a); }
void fr_mul(fp* r, const fp* a, const fp* b) { mr_mul(r, a, b); }
void fr_inv(fp* r, const fp* a) { m_inv(&MR, r, a); }
void fr_pow_u64(fp* r, const fp* a, uint64_t e) { const uint64_t ee[4] = {e, 0, 0, 0}; m_pow(&MR, r, a, ee); }
void fr_from_u64(fp* r, uint64_t x) { const uint64_t w[4] = {x, 0, 0, 0}; m_from_raw(&MR, r, w); }
void fr_from_bytes_mod_order(fp* r, const uint8_t b[32]) { uint64_t w[4]; le_to_limbs(w, b); m_from_raw(&MR, r, w); }
void fr_from_bytes_wide(fp* r, const uint8_t b[64]) {
    fp lo, hi; fr_from_bytes_mod_order(&lo, b); fr_from_bytes_mod_order(&hi, b + 32);
    m_mul(&MR, &hi, &hi, &MR.r2);                            /* hi * 2^256 */
    m_add(&MR, r, &lo, &hi);
}
void fr_to_raw(uint64_t w[4], const fp* a) { m_to_raw(&MR, w, a); }
void fr_to_bytes(uint8_t b[32], const fp* a) { uint64_t w[4]; m_to_raw(&MR, w, a); limbs_to_le(b, w); }
void fr_root_of_unity(fp* r, uint32_t m) {
    /* (r - 1) / m for a power of two m <= 2^28 */
    uint64_t e[4]; const uint64_t one[4] = {1, 0, 0, 0};
    sub4(e, MR.p, one);
    int sh = 0; while ((1u << sh) < m) sh++;
    for (int i = 0; i < 4; i++) e[i] = sh ? ((e[i] >> sh) | (i < 3 ? e[i + 1] << (64 - sh) : 0)) : e[i];
    fp g; fr_from_u64(&g, 5);
    m_pow(&MR, r, &g, e);
}

/* ---------------------------------------------------------------- Fq2 */
static void f2_add(fq2* r, const fq2* a, const fq2* b) { fq_add(&r->c0, &a->c0, &b->c0); fq_add(&r->c1, &a->c1, &b->c1); }
static void f2_sub(fq2* r, const fq2* a, const fq2* b) { fq_sub(&r->c0, &a->c0, &b->c0); fq_sub(&r->c1, &a->c1, &b->c1); }
static void f2_mul(fq2* r, const fq2* a, const fq2* b) {
    fp t0, t1, s0, s1, m;
    fq_mul(&t0, &a->c0, &b->c0); fq_mul(&t1, &a->c1, &b->c1);
    fq_add(&s0, &a->c0, &a->c1); fq_add(&s1, &b->c0, &b->c1); fq_mul(&m, &s0, &s1);
    fq_sub(&r->c0, &t0, &t1);
    fq_sub(&m, &m, &t0); fq_sub(&r->c1, &m, &t1);
}
static void f2_inv(fq2* r, const fq2* a) {
    fp n, t; fq_mul(&n, &a->c0, &a->c0); fq_mul(&t, &a->c1, &a->c1); fq_add(&n, &n, &t); fq_inv(&n, &n);
    fq_mul(&r->c0, &a->c0, &n); fq_mul(&t, &a->c1, &n); fq_neg(&r->c1, &t);
}
static int f2_is_zero(const fq2* a) { return fq_is_zero(&a->c0) && fq_is_zero(&a->c1); }

/* ---------------------------------------------------------------- curves: the same Jacobian formulas over Fq and Fq2 */
#define F fp
#define F_ADD fq_add
#define F_SUB fq_sub
#define F_MUL fq_mul
#define F_INV fq_inv
#define F_ISZERO fq_is_zero
#define AFF g1a
#define JAC g1j
#define FN(name) g1j_##name
#define MSM_NAME g1_msm
#include "bn254_curve.inc"
#undef F
#undef F_ADD
#undef F_SUB
#undef F_MUL
#undef F_INV
#undef F_ISZERO
#undef AFF
#undef JAC
#undef FN
#undef MSM_NAME

#define F fq2
#define F_ADD f2_add
#define F_SUB f2_sub
#define F_MUL f2_mul
#define F_INV f2_inv
#define F_ISZERO f2_is_zero
#define AFF g2a
#define JAC g2j
#define FN(name) g2j_##name
#define MSM_NAME g2_msm
#include "bn254_curve.inc"

void g1j_neg(g1j* r, const g1j* p) { *r = *p; fq_neg(&r->Y, &p->Y); }

/* ---------------------------------------------------------------- ark-serialize, uncompressed */
int g1_parse(g1a* r, const uint8_t b[64]) {
    uint8_t t[64]; memcpy(t, b, 64);
    const uint8_t flags = t[63] & 0xC0; t[63] &= 0x3F;
    memset(r, 0, sizeof *r);
    if (flags & 0x40) { r->inf = 1; return 1; }
    return fq_from_bytes(&r->x, t) && fq_from_bytes(&r->y, t + 32);
}
void g1_serialize(uint8_t b[64], const g1a* p) {
    memset(b, 0, 64);
    if (p->inf) { b[63] |= 0x40; return; }
    fq_to_bytes(b, &p->x); fq_to_bytes(b + 32, &p->y);
    if (fq_lex_larger(&p->y)) b[63] |= 0x80;
}
int g2_parse(g2a* r, const uint8_t b[128]) {
    uint8_t t[128]; memcpy(t, b, 128);
    const uint8_t flags = t[127] & 0xC0; t[127] &= 0x3F;
    memset(r, 0, sizeof *r);
    if (flags & 0x40) { r->inf = 1; return 1; }
    return fq_from_bytes(&r->x.c0, t) && fq_from_bytes(&r->x.c1, t + 32) && fq_from_bytes(&r->y.c0, t + 64) && fq_from_bytes(&r->y.c1, t + 96);
}
void g2_serialize(uint8_t b[128], const g2a* p) {
    memset(b, 0, 128);
    if (p->inf) { b[127] |= 0x40; return; }
    fq_to_bytes(b, &p->x.c0); fq_to_bytes(b + 32, &p->x.c1); fq_to_bytes(b + 64, &p->y.c0); fq_to_bytes(b + 96, &p->y.c1);
    /* ark_ff QuadExtField ordering: c1 first, then c0 */
    const int larger = fq_is_zero(&p->y.c1) ? fq_lex_larger(&p->y.c0) : fq_lex_larger(&p->y.c1);
    if (larger) b[127] |= 0x80;
}
