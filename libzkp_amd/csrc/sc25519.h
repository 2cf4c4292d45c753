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

// ---- modular inverse by Bernstein-Yang "safegcd" divsteps (constant-time, branch-free; 20 x 30 divsteps cover any
// 256-bit input, bound 590).  ~10k VALU instructions against ~78k for the Fermat ladder: the per-proof challenge
// inversions (y, u_1..u_6) sit on the serial critical path of every inner-product round.
// Signed 30-bit limbs, value = sum v[i] * 2^(30 i).
struct sc_s30 { int32_t v[9]; };
#define ZKP_SC_L30(i) ((i) == 0 ? 0x1cf5d3ed : (i) == 1 ? 0x20498c69 : (i) == 2 ? 0x2f79cd65 : (i) == 3 ? 0x37be77a8 : (i) == 4 ? 0x14 : (i) == 8 ? 0x1000 : 0)
#define ZKP_SC_L_INV30 0x2dab81e5u

struct sc_trans2x2 { int32_t u, v, q, r; };

ZKP_HD inline int32_t sc_divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, sc_trans2x2& t) {
    uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
    for (int i = 0; i < 30; i++) {
        uint32_t mask1 = (uint32_t)(zeta >> 31);
        const uint32_t mask2 = 0u - (g & 1u);
        const uint32_t x = (f ^ mask1) - mask1, y = (u ^ mask1) - mask1, z = (v ^ mask1) - mask1;
        g += x & mask2; q += y & mask2; r += z & mask2;
        mask1 &= mask2;
        zeta = (int32_t)((uint32_t)zeta ^ mask1) - 1;
        f += g & mask1; u += q & mask1; v += r & mask1;
        g >>= 1; u <<= 1; v <<= 1;
    }
    t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
    return zeta;
}
// (d, e) <- t * (d, e) / 2^30 mod l, keeping d, e in (-2l, l)
ZKP_HD inline void sc_update_de_30(sc_s30& d, sc_s30& e, const sc_trans2x2& t) {
    const int32_t M30 = (int32_t)(0xffffffffu >> 2);
    const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
    const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    int32_t di = d.v[0], ei = e.v[0];
    int64_t cd = (int64_t)u * di + (int64_t)v * ei, ce = (int64_t)q * di + (int64_t)r * ei;
    md -= (int32_t)((ZKP_SC_L_INV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((ZKP_SC_L_INV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cd += (int64_t)ZKP_SC_L30(0) * md; ce += (int64_t)ZKP_SC_L30(0) * me;
    cd >>= 30; ce >>= 30;
    ZKP_UNROLL for (int i = 1; i < 9; i++) {
        di = d.v[i]; ei = e.v[i];
        cd += (int64_t)u * di + (int64_t)v * ei; ce += (int64_t)q * di + (int64_t)r * ei;
        cd += (int64_t)ZKP_SC_L30(i) * md; ce += (int64_t)ZKP_SC_L30(i) * me;
        d.v[i - 1] = (int32_t)cd & M30; cd >>= 30;
        e.v[i - 1] = (int32_t)ce & M30; ce >>= 30;
    }
    d.v[8] = (int32_t)cd; e.v[8] = (int32_t)ce;
}
// (f, g) <- t * (f, g) / 2^30 (exact)
ZKP_HD inline void sc_update_fg_30(sc_s30& f, sc_s30& g, const sc_trans2x2& t) {
    const int32_t M30 = (int32_t)(0xffffffffu >> 2);
    const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
    int32_t fi = f.v[0], gi = g.v[0];
    int64_t cf = (int64_t)u * fi + (int64_t)v * gi, cg = (int64_t)q * fi + (int64_t)r * gi;
    cf >>= 30; cg >>= 30;
    ZKP_UNROLL for (int i = 1; i < 9; i++) {
        fi = f.v[i]; gi = g.v[i];
        cf += (int64_t)u * fi + (int64_t)v * gi; cg += (int64_t)q * fi + (int64_t)r * gi;
        f.v[i - 1] = (int32_t)cf & M30; cf >>= 30;
        g.v[i - 1] = (int32_t)cg & M30; cg >>= 30;
    }
    f.v[8] = (int32_t)cf; g.v[8] = (int32_t)cg;
}
// raw words (any value < l) -> raw words of its inverse mod l (0 -> 0)
ZKP_HD inline sc sc_modinv_raw(const sc& x) {
    const int32_t M30 = (int32_t)(0xffffffffu >> 2);
    sc_s30 d, e, f, g;
    ZKP_UNROLL for (int i = 0; i < 9; i++) { d.v[i] = 0; e.v[i] = 0; f.v[i] = ZKP_SC_L30(i); }
    e.v[0] = 1;
    // 8 x 32 -> 9 x 30
    ZKP_UNROLL for (int i = 0; i < 9; i++) {
        const int bit = 30 * i, w = bit >> 5, sh = bit & 31;
        uint32_t val = x.v[w] >> sh;
        if (sh > 2 && w + 1 < 8) val |= x.v[w + 1] << (32 - sh);
        g.v[i] = (int32_t)(val & (uint32_t)M30);
    }
    int32_t zeta = -1;
    for (int it = 0; it < 20; it++) {
        sc_trans2x2 t;
        zeta = sc_divsteps_30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
        sc_update_de_30(d, e, t);
        sc_update_fg_30(f, g, t);
    }
    // now g = 0 and f = +-1; inverse = sign(f) * d, d in (-2l, l)
    const int32_t neg = f.v[8] >> 31;                       // all ones if f = -1
    int64_t c = 0; int32_t lim[9];
    ZKP_UNROLL for (int i = 0; i < 9; i++) { c += (int64_t)((d.v[i] ^ neg) - neg); lim[i] = (int32_t)c & M30; c >>= 30; }
    lim[8] |= (int32_t)((uint32_t)c << 30);                 // keep the sign in the top limb (value in (-2l, 2l))
    // add l while negative (at most twice), then subtract l if >= l
    for (int rep = 0; rep < 2; rep++) {
        const int32_t m = lim[8] >> 31;
        int64_t cc = 0;
        ZKP_UNROLL for (int i = 0; i < 8; i++) { cc += (int64_t)lim[i] + (ZKP_SC_L30(i) & m); lim[i] = (int32_t)cc & M30; cc >>= 30; }
        lim[8] = (int32_t)(cc + lim[8] + (ZKP_SC_L30(8) & m));
    }
    {
        int64_t cc = 0; int32_t sub[9];
        ZKP_UNROLL for (int i = 0; i < 8; i++) { cc += (int64_t)lim[i] - ZKP_SC_L30(i); sub[i] = (int32_t)cc & M30; cc >>= 30; }
        sub[8] = (int32_t)(cc + lim[8] - ZKP_SC_L30(8));
        const bool ge = sub[8] >= 0;
        ZKP_UNROLL for (int i = 0; i < 9; i++) lim[i] = ge ? sub[i] : lim[i];
    }
    // 9 x 30 -> 8 x 32
    sc r;
    ZKP_UNROLL for (int w = 0; w < 8; w++) {
        const int bit = 32 * w, i = bit / 30, sh = bit % 30;
        uint32_t val = (uint32_t)lim[i] >> sh;
        val |= (uint32_t)lim[i + 1] << (30 - sh);
        if (30 - sh + 30 < 32 && i + 2 < 9) val |= (uint32_t)lim[i + 2] << (60 - sh);
        r.v[w] = val;
    }
    return r;
}
// inverse in the Montgomery domain: (aR)^-1 * R^3 / R = a^-1 R
ZKP_HD inline sc sc_invert(const sc& a) { return sc_montmul(sc_modinv_raw(a), sc_R3()); }

// a^(l-2) by a fixed 4-bit window ladder (kept as the independent cross-check of sc_invert in the tests)
ZKP_HD inline sc sc_invert_fermat(const sc& a) {
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
            for (int k = 2; k < 16; k++) m = (d == (uint32_t)k) ? tbl[k] : m;
            acc = sc_montmul(acc, m);
        }
    }
    return acc;
}

// signed radix-1024 recoding of a canonical raw scalar (< 2^253): 26 digits in [-511, 512], two 16-bit digits per word
// (13 words).  Used by the Bulletproofs fixed-base tables (512 entries per window, 26 windows).
ZKP_HD inline void sc_recode_signed1024(uint32_t packed[13], const sc& raw) {
    uint32_t carry = 0;
    ZKP_UNROLL for (int j = 0; j < 26; j++) {
        const int bit = 10 * j, wd = bit >> 5, sh = bit & 31;
        uint32_t x = wd < 8 ? raw.v[wd] >> sh : 0u;
        if (sh > 22 && wd + 1 < 8) x |= raw.v[wd + 1] << (32 - sh);
        uint32_t d = (x & 0x3ffu) + carry;                          // 0..1024
        carry = d > 512u ? 1u : 0u;                                 // digit = d - 1024*carry in [-511, 512]
        d = (d - (carry << 10)) & 0xffffu;
        if ((j & 1) == 0) packed[j >> 1] = d; else packed[j >> 1] |= d << 16;
    }
}

// signed radix-2^16 digits of a canonical raw scalar (< 2^253): 16 digits in [-32768, 32767], digit j in the half (j & 1) of word j / 2.
// A digit of 32768 becomes -32768 with a carry; |digit| <= 32768 = entries per window of the HBM-resident tables (edg.h); the top digit (bits 240..252 plus the carry)
// never overflows.  A 64-bit value occupies digits 0..4 (five windows).
ZKP_HD inline void sc_recode_signed65536(uint32_t packed[8], const sc& raw) {
    uint32_t carry = 0;
    ZKP_UNROLL for (int k = 0; k < 8; k++) {
        const uint32_t lo = (raw.v[k] & 0xffffu) + carry;               // 0 .. 65536
        carry = lo > 32767u ? 1u : 0u;
        const uint32_t hi = (raw.v[k] >> 16) + carry;
        carry = hi > 32767u ? 1u : 0u;
        packed[k] = (lo & 0xffffu) | (hi << 16);                        // (hi == 65536 wraps to 0 with the carry set)
    }
}

// signed radix-2^WB recoding of a canonical raw scalar: ND digits in [-(2^(WB-1) - 1), 2^(WB-1)], two 16-bit digits per word
// (WB <= 15; WB * ND must cover the scalar's bit length plus the carry).  WB = 10, ND = 26 is sc_recode_signed1024.
template <int WB, int ND> ZKP_HD inline void sc_recode_signed(uint32_t* packed, const sc& raw) {
    uint32_t carry = 0;
    ZKP_UNROLL for (int j = 0; j < ND; j++) {
        const int bit = WB * j, wd = bit >> 5, sh = bit & 31;
        uint32_t x = wd < 8 ? raw.v[wd] >> sh : 0u;
        if (sh + WB > 32 && wd + 1 < 8) x |= raw.v[wd + 1] << (32 - sh);
        uint32_t d = (x & ((1u << WB) - 1u)) + carry;              // 0 .. 2^WB
        carry = d > (1u << (WB - 1)) ? 1u : 0u;                    // digit = d - 2^WB * carry
        d = (d - (carry << WB)) & 0xffffu;
        if ((j & 1) == 0) packed[j >> 1] = d; else packed[j >> 1] |= d << 16;
    }
}

}  // namespace zkp
