// BN254 base field Fq for gfx950 in UNSATURATED Montgomery form: ten 26-bit limbs (R = 2^260), product-scanning
// multiplication with the Montgomery reduction interleaved column by column.  Every column sum (<= 10 products of
// 29-bit x 29-bit limbs + 10 reduction products + carry) fits one 64-bit accumulator, so the compiler emits a clean
// v_mad_u64_u32 chain -- the saturated 8x32 CIOS form spent 2.7 v_mov/v_lshl_add_u64 per v_mad on carry shuffling
// (measured: 3848 v_mov + 1539 v_lshl_add_u64 vs 1415 v_mad per G1 mixed addition).
// Replaces ark-ff's Fq for ark-bn254 (used under /root/reference/src/backend/snark.rs:4-12,364,442).
//
// Bounds vocabulary (checked by tests/test_fq_bounds.py):
//   carried : limbs 0..8 < 2^26 (limb 9 holds the rest);   safe : carried and value < 4p
//   fq_mul / fq_sq accept limbs < 2^29 and return a carried value < A*B/(84.6 p) + p   (p/R = 1/84.6)
//   fq_add_l / fq_sub_k* are limb-wise (no carries); fq_sub_kN adds N*p in a borrowed form whose low limbs are >= 2^(24+log2 N)
//   fq_reduce_weak brings any value < 2^260 to [0, 3p), carried.
#pragma once
#include "bn254_fp.h"

namespace zkp {

struct fq { uint32_t v[10]; };
#define ZKP_FQ_MASK 0x3ffffffu
ZKP_HD constexpr uint32_t fq_pl(int i) { constexpr uint32_t m[10] = {0x7cfd47u, 0x2305b6u, 0xa8d3c2u, 0x245a1c7u, 0x197816au, 0x605617u, 0x1045b68u, 0x280a6e1u, 0x272e131u, 0xc1913u}; return m[i]; }
ZKP_HD constexpr uint32_t fq_one_l(int i) { constexpr uint32_t m[10] = {0x2fce4b4u, 0x82203du, 0x9a8455u, 0x126eaa6u, 0x2498908u, 0x63c052u, 0x29201d8u, 0x1c93e16u, 0x24e1bb7u, 0x7c590u}; return m[i]; }
ZKP_HD constexpr uint32_t fq_r2_l(int i) { constexpr uint32_t m[10] = {0x166eb04u, 0x22a0746u, 0x16b86u, 0x1865406u, 0x98e615u, 0x2d3e263u, 0x1531600u, 0x265a6ffu, 0x1a30d3au, 0x2a11au}; return m[i]; }
ZKP_HD constexpr uint32_t fq_k4(int i) { constexpr uint32_t m[10] = {0x5f3f51cu, 0x48c16d7u, 0x6a34f07u, 0x516871bu, 0x65e05a9u, 0x581585cu, 0x4116d9fu, 0x6029b84u, 0x5cb84c5u, 0x30644du}; return m[i]; }
ZKP_HD constexpr uint32_t fq_k8(int i) { constexpr uint32_t m[10] = {0xbe7ea38u, 0x9182daeu, 0x9469e0eu, 0xa2d0e37u, 0x8bc0b52u, 0xb02b0b9u, 0x822db3eu, 0x8053708u, 0xb97098bu, 0x60c89au}; return m[i]; }
ZKP_HD constexpr uint32_t fq_k16(int i) { constexpr uint32_t m[10] = {0x13cfd470u, 0x12305b5du, 0x128d3c1cu, 0x105a1c6eu, 0x117816a5u, 0x12056172u, 0x1045b67du, 0x100a6e10u, 0x132e1316u, 0xc19135u}; return m[i]; }
#define ZKP_FQ_N0 0x866389u          // -p^-1 mod 2^26
#define ZKP_FQ_RECIP 5417u           // floor(2^266 / p)

ZKP_HD inline fq fq_zero() { fq r; ZKP_UNROLL for (int i = 0; i < 10; i++) r.v[i] = 0; return r; }
ZKP_HD inline fq fq_one() { fq r; ZKP_UNROLL for (int i = 0; i < 10; i++) r.v[i] = fq_one_l(i); return r; }

// a * b / 2^260 mod p
ZKP_HD inline fq fq_mul(const fq& a, const fq& b) {
    uint32_t m[10]; fq r; uint64_t acc = 0;
    ZKP_UNROLL for (int i = 0; i < 10; i++) {
        ZKP_UNROLL for (int j = 0; j <= i; j++) acc += (uint64_t)a.v[j] * b.v[i - j];
        ZKP_UNROLL for (int j = 0; j < i; j++) acc += (uint64_t)m[j] * fq_pl(i - j);
        m[i] = ((uint32_t)acc * ZKP_FQ_N0) & ZKP_FQ_MASK;
        acc += (uint64_t)m[i] * fq_pl(0);
        acc >>= 26;
    }
    ZKP_UNROLL for (int i = 10; i < 19; i++) {
        ZKP_UNROLL for (int j = i - 9; j < 10; j++) acc += (uint64_t)a.v[j] * b.v[i - j];
        ZKP_UNROLL for (int j = i - 9; j < 10; j++) acc += (uint64_t)m[j] * fq_pl(i - j);
        r.v[i - 10] = (uint32_t)acc & ZKP_FQ_MASK;
        acc >>= 26;
    }
    r.v[9] = (uint32_t)acc;
    return r;
}
// Squaring: the product columns are symmetric, so each pair a_j a_(i-j) is formed once with a doubled limb (55 products
// instead of 100; the 100 reduction products stay).  Same input / output contract as fq_mul (limbs < 2^29 doubled still
// fit 32 bits, column sums are term for term those of fq_mul(a, a)).
ZKP_HD inline fq fq_sq(const fq& a) {
    uint32_t m[10], d[10]; fq r; uint64_t acc = 0;
    ZKP_UNROLL for (int i = 0; i < 10; i++) d[i] = a.v[i] << 1;
    ZKP_UNROLL for (int i = 0; i < 10; i++) {
        ZKP_UNROLL for (int j = 0; 2 * j < i; j++) acc += (uint64_t)d[j] * a.v[i - j];
        if ((i & 1) == 0) acc += (uint64_t)a.v[i / 2] * a.v[i / 2];
        ZKP_UNROLL for (int j = 0; j < i; j++) acc += (uint64_t)m[j] * fq_pl(i - j);
        m[i] = ((uint32_t)acc * ZKP_FQ_N0) & ZKP_FQ_MASK;
        acc += (uint64_t)m[i] * fq_pl(0);
        acc >>= 26;
    }
    ZKP_UNROLL for (int i = 10; i < 19; i++) {
        ZKP_UNROLL for (int j = i - 9; 2 * j < i; j++) acc += (uint64_t)d[j] * a.v[i - j];
        if ((i & 1) == 0) acc += (uint64_t)a.v[i / 2] * a.v[i / 2];
        ZKP_UNROLL for (int j = i - 9; j < 10; j++) acc += (uint64_t)m[j] * fq_pl(i - j);
        r.v[i - 10] = (uint32_t)acc & ZKP_FQ_MASK;
        acc >>= 26;
    }
    r.v[9] = (uint32_t)acc;
    return r;
}

// (a * b + c * d) / 2^260 mod p with ONE Montgomery reduction: both products are accumulated into the same columns (200 limb
// products), then reduced once (100) -- 300 multiply-adds where two fq_mul and an addition take 400.  Inputs: limbs < 2^28
// each (column sums of 20 products of < 2^56 plus the reduction terms stay below 2^61); output carried, value < (A*B + C*D) /
// (84.6 p) + p.
ZKP_HD inline fq fq_mul_add2(const fq& a, const fq& b, const fq& c, const fq& d) {
    uint32_t m[10]; fq r; uint64_t acc = 0;
    ZKP_UNROLL for (int i = 0; i < 10; i++) {
        ZKP_UNROLL for (int j = 0; j <= i; j++) { acc += (uint64_t)a.v[j] * b.v[i - j]; acc += (uint64_t)c.v[j] * d.v[i - j]; }
        ZKP_UNROLL for (int j = 0; j < i; j++) acc += (uint64_t)m[j] * fq_pl(i - j);
        m[i] = ((uint32_t)acc * ZKP_FQ_N0) & ZKP_FQ_MASK;
        acc += (uint64_t)m[i] * fq_pl(0);
        acc >>= 26;
    }
    ZKP_UNROLL for (int i = 10; i < 19; i++) {
        ZKP_UNROLL for (int j = i - 9; j < 10; j++) { acc += (uint64_t)a.v[j] * b.v[i - j]; acc += (uint64_t)c.v[j] * d.v[i - j]; }
        ZKP_UNROLL for (int j = i - 9; j < 10; j++) acc += (uint64_t)m[j] * fq_pl(i - j);
        r.v[i - 10] = (uint32_t)acc & ZKP_FQ_MASK;
        acc >>= 26;
    }
    r.v[9] = (uint32_t)acc;
    return r;
}

// limb-wise (lazy) operations
ZKP_HD inline fq fq_add_l(const fq& a, const fq& b) { fq r; ZKP_UNROLL for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
ZKP_HD inline fq fq_dbl_l(const fq& a) { fq r; ZKP_UNROLL for (int i = 0; i < 10; i++) r.v[i] = a.v[i] << 1; return r; }
ZKP_HD inline fq fq_sub_k4(const fq& a, const fq& b) { fq r; ZKP_UNROLL for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + fq_k4(i) - b.v[i]; return r; }     // b: low limbs <= 2^26, value < 4p
ZKP_HD inline fq fq_sub_k8(const fq& a, const fq& b) { fq r; ZKP_UNROLL for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + fq_k8(i) - b.v[i]; return r; }     // b: low limbs <= 2^27, value < 8p
ZKP_HD inline fq fq_sub_k16(const fq& a, const fq& b) { fq r; ZKP_UNROLL for (int i = 0; i < 10; i++) r.v[i] = a.v[i] + fq_k16(i) - b.v[i]; return r; }   // b: low limbs <= 2^28, value < 16p

// limb normalisation (value unchanged); input limbs < 2^32, value < 2^260
ZKP_HD inline fq fq_carry(const fq& a) {
    fq r; uint32_t c = 0;
    ZKP_UNROLL for (int i = 0; i < 9; i++) { const uint32_t t = a.v[i] + c; r.v[i] = t & ZKP_FQ_MASK; c = t >> 26; }
    r.v[9] = a.v[9] + c;
    return r;
}
// any value < 2^260 (limbs < 2^31) -> [0, 3p), carried: subtract q*p with q = floor(top * floor(2^266/p) / 2^32) <= floor(x/p)
ZKP_HD inline fq fq_reduce_weak(const fq& a) {
    const fq c = fq_carry(a);
    const uint32_t q = (uint32_t)(((uint64_t)c.v[9] * ZKP_FQ_RECIP) >> 32);
    fq r; int64_t acc = 0;
    ZKP_UNROLL for (int i = 0; i < 9; i++) {
        acc += (int64_t)c.v[i] - (int64_t)((uint64_t)q * fq_pl(i));
        r.v[i] = (uint32_t)acc & ZKP_FQ_MASK; acc >>= 26;
    }
    r.v[9] = (uint32_t)(acc + (int64_t)c.v[9] - (int64_t)((uint64_t)q * fq_pl(9)));
    return r;
}
// "safe" (always reduced) operations for code that is not on the hot path
ZKP_HD inline fq fq_add(const fq& a, const fq& b) { return fq_reduce_weak(fq_add_l(a, b)); }     // inputs < 2^31 limb sums
ZKP_HD inline fq fq_sub(const fq& a, const fq& b) { return fq_reduce_weak(fq_sub_k8(a, b)); }    // b safe
ZKP_HD inline fq fq_neg(const fq& a) { return fq_sub(fq_zero(), a); }
ZKP_HD inline fq fq_dbl(const fq& a) { return fq_reduce_weak(fq_dbl_l(a)); }
ZKP_HD inline fq fq_select(bool c, const fq& a, const fq& b) { fq r; ZKP_UNROLL for (int i = 0; i < 10; i++) r.v[i] = c ? a.v[i] : b.v[i]; return r; }

// raw little-endian 8 x 32-bit words <-> Montgomery limbs
ZKP_HD inline fq fq_unpack(const uint32_t w[8]) {    // integer value of the words, 26-bit limbs (no Montgomery factor)
    fq r;
    ZKP_UNROLL for (int i = 0; i < 10; i++) {
        const int bit = 26 * i, wd = bit >> 5, sh = bit & 31;
        uint32_t x = w[wd] >> sh;
        if (sh > 6 && wd + 1 < 8) x |= w[wd + 1] << (32 - sh);
        r.v[i] = i < 9 ? (x & ZKP_FQ_MASK) : x;
    }
    return r;
}
ZKP_HD inline fq fq_from_raw(const uint32_t w[8]) {   // any raw < 2^256 -> Montgomery form of raw mod p (< 2p)
    fq r2; ZKP_UNROLL for (int i = 0; i < 10; i++) r2.v[i] = fq_r2_l(i);
    return fq_mul(fq_unpack(w), r2);
}
ZKP_HD inline void fq_to_raw(uint32_t w[8], const fq& a) {   // canonical value in [0, p)
    fq one; ZKP_UNROLL for (int i = 0; i < 10; i++) one.v[i] = i == 0 ? 1u : 0u;
    fq x = fq_mul(fq_carry(a), one);                  // < a/R + p < 2p for any a < 2^260, carried
    // conditional subtraction of p
    uint32_t d[10]; int64_t acc = 0;
    ZKP_UNROLL for (int i = 0; i < 9; i++) { acc += (int64_t)x.v[i] - (int64_t)fq_pl(i); d[i] = (uint32_t)acc & ZKP_FQ_MASK; acc >>= 26; }
    acc += (int64_t)x.v[9] - (int64_t)fq_pl(9);
    const bool neg = acc < 0; d[9] = (uint32_t)acc;
    ZKP_UNROLL for (int i = 0; i < 10; i++) x.v[i] = neg ? x.v[i] : d[i];
    // 10 x 26 -> 8 x 32
    ZKP_UNROLL for (int wd = 0; wd < 8; wd++) {
        const int bit = 32 * wd, i = bit / 26, sh = bit % 26;
        uint32_t val = x.v[i] >> sh;
        val |= x.v[i + 1] << (26 - sh);
        if (26 - sh + 26 < 32 && i + 2 < 10) val |= x.v[i + 2] << (52 - sh);
        w[wd] = val;
    }
}
ZKP_HD inline fq fq_from_u64(uint64_t x) { const uint32_t w[8] = {(uint32_t)x, (uint32_t)(x >> 32), 0, 0, 0, 0, 0, 0}; return fq_from_raw(w); }
ZKP_HD inline bool fq_is_zero(const fq& a) { uint32_t w[8]; fq_to_raw(w, a); uint32_t o = 0; ZKP_UNROLL for (int i = 0; i < 8; i++) o |= w[i]; return o == 0; }
ZKP_HD inline bool fq_eq(const fq& a, const fq& b) { return fq_is_zero(fq_sub(a, fq_reduce_weak(b))); }
// a^(p-2), Fermat ladder (a handful per proof: affine conversion of the proof points; table construction)
ZKP_HD inline fq fq_inv(const fq& a) {
    fq acc = fq_one();
    const fq base = fq_reduce_weak(a);
    for (int i = 255; i >= 0; i--) {
        acc = fq_sq(acc);
        uint32_t w = FqParams::mod(i >> 5);
        if ((i >> 5) == 0) w -= 2;
        if ((w >> (i & 31)) & 1u) acc = fq_mul(acc, base);
    }
    return acc;
}

}  // namespace zkp
