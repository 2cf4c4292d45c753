// Scalars modulo l = 2^252 + 27742317777372353535851937790883648493 for gfx950: eight 32-bit words,
// kept in Montgomery form (x*2^256 mod l, always fully reduced) between kernels so that one product is
// one CIOS pass of v_mad_u64_u32.  Replaces curve25519-dalek's Scalar as used at
// /root/reference/src/backend/bulletproofs.rs:5,86 (from_bytes_mod_order) and inside bulletproofs' prover.
#pragma once
#include "zkp_common.h"

namespace zkp {

struct sc { uint32_t v[8]; };   // Montgomery form unless a function says "raw"

#define ZKP_SC_L(i) ((i) == 0 ? 0x5cf5d3edu : (i) == 1 ? 0x5812631au : (i) == 2 ? 0xa2f79cd6u : (i) == 3 ? 0x14def9deu : (i) == 7 ? 0x10000000u : 0u)
#define ZKP_SC_N0 0x12547e1bu

ZKP_HD inline sc sc_words(uint32_t a, uint32_t b, uint32_t c, uint32_t d, uint32_t e, uint32_t f, uint32_t g, uint32_t h) {
    sc r; r.v[0] = a; r.v[1] = b; r.v[2] = c; r.v[3] = d; r.v[4] = e; r.v[5] = f; r.v[6] = g; r.v[7] = h; return r;
}
ZKP_HD inline sc sc_zero() { return sc_words(0, 0, 0, 0, 0, 0, 0, 0); }
ZKP_HD inline sc sc_raw_one() { return sc_words(1, 0, 0, 0, 0, 0, 0, 0); }
ZKP_HD inline sc sc_one() { return sc_words(0x8d98951du, 0xd6ec3174u, 0x737dcf70u, 0xc6ef5bf4u, 0xfffffffeu, 0xffffffffu, 0xffffffffu, 0x0fffffffu); }  // R mod l
ZKP_HD inline sc sc_R2() { return sc_words(0x449c0f01u, 0xa40611e3u, 0x68859347u, 0xd00e1ba7u, 0x17f5be65u, 0xceec73d2u, 0x7c309a3du, 0x0399411bu); }
ZKP_HD inline sc sc_R3() { return sc_words(0x7b83a2dbu, 0x2a9e4968u, 0xaef7f3ecu, 0x278324e6u, 0x04ec5b65u, 0x8065dc6cu, 0x3599cec7u, 0x0e530b77u); }

// r = t - l if t >= l else t   (t < 2l)
ZKP_HD inline sc sc_cond_sub_l(const uint32_t t[8]) {
    uint32_t d[8]; uint64_t br = 0;
    ZKP_UNROLL for (int i = 0; i < 8; i++) {
        uint64_t x = (uint64_t)t[i] - ZKP_SC_L(i) - br;
        d[i] = (uint32_t)x; br = (x >> 32) & 1;
    }
    sc r; ZKP_UNROLL for (int i = 0; i < 8; i++) r.v[i] = br ? t[i] : d[i];
    return r;
}

// a*b*2^-256 mod l ; requires a < 2^256 and b < l ; result < l
ZKP_HD inline sc sc_montmul(const sc& a, const sc& b) {
    uint32_t t[10];
    ZKP_UNROLL for (int i = 0; i < 10; i++) t[i] = 0;
    ZKP_UNROLL for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
        ZKP_UNROLL for (int j = 0; j < 8; j++) { c += (uint64_t)a.v[j] * b.v[i] + t[j]; t[j] = (uint32_t)c; c >>= 32; }
        c += t[8]; t[8] = (uint32_t)c; t[9] = (uint32_t)(c >> 32);
        const uint32_t m = t[0] * ZKP_SC_N0;
        c = (uint64_t)m * ZKP_SC_L(0) + t[0]; c >>= 32;
        ZKP_UNROLL for (int j = 1; j < 8; j++) { c += (uint64_t)m * ZKP_SC_L(j) + t[j]; t[j - 1] = (uint32_t)c; c >>= 32; }
        c += t[8]; t[7] = (uint32_t)c; t[8] = t[9] + (uint32_t)(c >> 32);
    }
    return sc_cond_sub_l(t);
}

ZKP_HD inline sc sc_mul(const sc& a, const sc& b) { return sc_montmul(a, b); }

ZKP_HD inline sc sc_add(const sc& a, const sc& b) {
    uint32_t t[8]; uint64_t c = 0;
    ZKP_UNROLL for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + b.v[i]; t[i] = (uint32_t)c; c >>= 32; }
    return sc_cond_sub_l(t);
}
ZKP_HD inline sc sc_sub(const sc& a, const sc& b) {
    uint32_t d[8]; uint64_t br = 0;
    ZKP_UNROLL for (int i = 0; i < 8; i++) { uint64_t x = (uint64_t)a.v[i] - b.v[i] - br; d[i] = (uint32_t)x; br = (x >> 32) & 1; }
    uint64_t c = 0; sc r;
    ZKP_UNROLL for (int i = 0; i < 8; i++) { c += (uint64_t)d[i] + (br ? ZKP_SC_L(i) : 0u); r.v[i] = (uint32_t)c; c >>= 32; }
    return r;
}
ZKP_HD inline sc sc_neg(const sc& a) { return sc_sub(sc_zero(), a); }
ZKP_HD inline sc sc_muladd(const sc& a, const sc& b, const sc& c) { return sc_add(sc_montmul(a, b), c); }

// conversions; "raw" = plain little-endian words
ZKP_HD inline sc sc_from_raw256(const sc& raw) { return sc_montmul(raw, sc_R2()); }          // any raw < 2^256 -> Montgomery form of raw mod l
ZKP_HD inline sc sc_to_raw(const sc& a) { return sc_montmul(a, sc_raw_one()); }              // canonical value words
ZKP_HD inline sc sc_from_u64(uint64_t x) { return sc_from_raw256(sc_words((uint32_t)x, (uint32_t)(x >> 32), 0, 0, 0, 0, 0, 0)); }
ZKP_HD inline sc sc_from_wide(const uint32_t w[16]) {   // Scalar::from_bytes_mod_order_wide
    sc lo, hi;
    ZKP_UNROLL for (int i = 0; i < 8; i++) { lo.v[i] = w[i]; hi.v[i] = w[8 + i]; }
    return sc_add(sc_montmul(lo, sc_R2()), sc_montmul(hi, sc_R3()));
}

// a^(l-2), fixed 4-bit window
ZKP_HD inline sc sc_invert(const sc& a) {
    sc tbl[16];
    tbl[0] = sc_one(); tbl[1] = a;
    for (int i = 2; i < 16; i++) tbl[i] = sc_montmul(tbl[i - 1], a);
    const uint32_t e[8] = {0x5cf5d3ebu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u};
    sc acc = tbl[1];  // top nibble of l-2 is 1
    for (int nib = 62; nib >= 0; nib--) {
        acc = sc_montmul(acc, acc); acc = sc_montmul(acc, acc); acc = sc_montmul(acc, acc); acc = sc_montmul(acc, acc);
        const uint32_t d = (e[nib >> 3] >> ((nib & 7) * 4)) & 15u;
        if (d) {
            sc m = tbl[1];
            for (int k = 2; k < 16; k++) m = (d == (uint32_t)k) ? tbl[k] : m;   // select without dynamic register indexing
            acc = sc_montmul(acc, m);
        }
    }
    return acc;
}

// signed radix-256 recoding of a canonical raw scalar (< 2^253): 32 digits in [-128, 127], packed 4 per word
ZKP_HD inline void sc_recode_signed256(uint32_t packed[8], const sc& raw) {
    uint32_t carry = 0;
    ZKP_UNROLL for (int w = 0; w < 8; w++) {
        uint32_t out = 0;
        ZKP_UNROLL for (int k = 0; k < 4; k++) {
            uint32_t d = ((raw.v[w] >> (8 * k)) & 0xffu) + carry;   // 0..256
            carry = d >= 128u ? 1u : 0u;                            // digit = d - 256*carry in [-128, 127]
            out |= (d & 0xffu) << (8 * k);
        }
        packed[w] = out;
    }
}

}  // namespace zkp
