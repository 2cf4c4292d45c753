// BN254 base field Fq in NINE 29-bit limbs (Montgomery form, R9 = 2^261): the representation of the Groth16 G1 MSM loop.
// A product is 81 + 81 multiply-adds and 17 carry shifts where bn254_fq.h's ten 26-bit limbs take 100 + 100 and 19 -- measured
// 171 against 143 G products/s on MI355X (tools/fq_microbench.hip, profiles/r02_fq_microbench.jsonl).  The price: a column of
// 9 + 9 products of 29-bit limbs leaves under two bits of a 64-bit accumulator, so operands must be CARRIED (every limb < 2^29);
// there are no limb-wise lazy additions here -- fq9_add / fq9_sub propagate carries -- while VALUES may run far above p
// (2^261 = 169.3 p).  Only the MSM loop lives in this form: key tables are converted when they are built, the accumulator when
// a chunk starts and ends (fq9_from_fq / fq9_to_fq); everything else stays in bn254_fq.h's form.
//
// Bounds vocabulary (tests/test_fq_bounds.py): carried9 = limbs 0..7 < 2^29 and the value < 2^261 (so limb 8 < 2^29 too).
//   fq9_mul / fq9_sq: carried9 in, carried9 out, value < a b / (169.28 p) + p
//   fq9_mul_add2    : carried9 in, value < (a b + c d) / (169.28 p) + p            (27 products of < 2^58 per column < 2^62.8)
//   fq9_sub_k<K>    : a - b + K p with b < K p; carried9 out, value < a + K p
#pragma once
#include "bn254_fq.h"

namespace zkp {

struct fq9 { uint32_t v[9]; };
#define ZKP_FQ9_MASK 0x1fffffffu
#define ZKP_FQ9_N0 0x4866389u         // -p^-1 mod 2^29
ZKP_HD constexpr uint32_t fq9_pl(int i) { constexpr uint32_t m[9] = {0x187cfd47u, 0x10460b6u, 0x1c72a34fu, 0x2d522d0u, 0x1585d978u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu}; return m[i]; }
ZKP_HD constexpr uint32_t fq9_k2(int i) { constexpr uint32_t m[9] = {0x10f9fa8eu, 0x208c16du, 0x18e5469eu, 0x5aa45a1u, 0xb0bb2f0u, 0x5b68181u, 0x14dc282u, 0x1cb84c68u, 0x60c89cu}; return m[i]; }
ZKP_HD constexpr uint32_t fq9_k4(int i) { constexpr uint32_t m[9] = {0x1f3f51cu, 0x41182dbu, 0x11ca8d3cu, 0xb548b43u, 0x161765e0u, 0xb6d0302u, 0x29b8504u, 0x197098d0u, 0xc19139u}; return m[i]; }
ZKP_HD constexpr uint32_t fq9_k8(int i) { constexpr uint32_t m[9] = {0x3e7ea38u, 0x82305b6u, 0x3951a78u, 0x16a91687u, 0xc2ecbc0u, 0x16da0605u, 0x5370a08u, 0x12e131a0u, 0x1832273u}; return m[i]; }
ZKP_HD constexpr uint32_t fq9_r10(int i) { constexpr uint32_t m[9] = {0x16fce4b4u, 0xa904407u, 0xa626a11u, 0x12109375u, 0x1014a498u, 0x100ec0c7u, 0x93e16a4u, 0x9c376eeu, 0x1f1642u}; return m[i]; }   // 2^260 mod p
ZKP_HD constexpr uint32_t fq9_k16(int i) { constexpr uint32_t m[9] = {0x7cfd470u, 0x10460b6cu, 0x72a34f0u, 0xd522d0eu, 0x185d9781u, 0xdb40c0au, 0xa6e1411u, 0x5c26340u, 0x30644e7u}; return m[i]; }
ZKP_HD constexpr uint32_t fq9_k32(int i) { constexpr uint32_t m[9] = {0xf9fa8e0u, 0x8c16d8u, 0xe5469e1u, 0x1aa45a1cu, 0x10bb2f02u, 0x1b681815u, 0x14dc2822u, 0xb84c680u, 0x60c89ceu}; return m[i]; }
template <int K> ZKP_HD constexpr uint32_t fq9_kp(int i) {
    static_assert(K == 1 || K == 2 || K == 4 || K == 8 || K == 16 || K == 32, "multiples of p held as constants");
    return K == 1 ? fq9_pl(i) : K == 2 ? fq9_k2(i) : K == 4 ? fq9_k4(i) : K == 8 ? fq9_k8(i) : K == 16 ? fq9_k16(i) : fq9_k32(i);
}

ZKP_HD inline fq9 fq9_zero() { fq9 r; ZKP_UNROLL for (int i = 0; i < 9; i++) r.v[i] = 0; return r; }

// a * b / 2^261 mod p
ZKP_HD inline fq9 fq9_mul(const fq9& a, const fq9& b) {
    uint32_t m[9]; fq9 r; uint64_t acc = 0;
    ZKP_UNROLL for (int i = 0; i < 9; i++) {
        ZKP_UNROLL for (int j = 0; j <= i; j++) acc += (uint64_t)a.v[j] * b.v[i - j];
        ZKP_UNROLL for (int j = 0; j < i; j++) acc += (uint64_t)m[j] * fq9_pl(i - j);
        m[i] = ((uint32_t)acc * ZKP_FQ9_N0) & ZKP_FQ9_MASK;
        acc += (uint64_t)m[i] * fq9_pl(0);
        acc >>= 29;
    }
    ZKP_UNROLL for (int i = 9; i < 17; i++) {
        ZKP_UNROLL for (int j = i - 8; j < 9; j++) acc += (uint64_t)a.v[j] * b.v[i - j];
        ZKP_UNROLL for (int j = i - 8; j < 9; j++) acc += (uint64_t)m[j] * fq9_pl(i - j);
        r.v[i - 9] = (uint32_t)acc & ZKP_FQ9_MASK;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
// a^2 / 2^261: 45 products with a doubled limb (< 2^30) + 81 reduction products; column sums term for term those of fq9_mul(a, a)
ZKP_HD inline fq9 fq9_sq(const fq9& a) {
    uint32_t m[9], d[9]; fq9 r; uint64_t acc = 0;
    ZKP_UNROLL for (int i = 0; i < 9; i++) d[i] = a.v[i] << 1;
    ZKP_UNROLL for (int i = 0; i < 9; i++) {
        ZKP_UNROLL for (int j = 0; 2 * j < i; j++) acc += (uint64_t)d[j] * a.v[i - j];
        if ((i & 1) == 0) acc += (uint64_t)a.v[i / 2] * a.v[i / 2];
        ZKP_UNROLL for (int j = 0; j < i; j++) acc += (uint64_t)m[j] * fq9_pl(i - j);
        m[i] = ((uint32_t)acc * ZKP_FQ9_N0) & ZKP_FQ9_MASK;
        acc += (uint64_t)m[i] * fq9_pl(0);
        acc >>= 29;
    }
    ZKP_UNROLL for (int i = 9; i < 17; i++) {
        ZKP_UNROLL for (int j = i - 8; 2 * j < i; j++) acc += (uint64_t)d[j] * a.v[i - j];
        if ((i & 1) == 0) acc += (uint64_t)a.v[i / 2] * a.v[i / 2];
        ZKP_UNROLL for (int j = i - 8; j < 9; j++) acc += (uint64_t)m[j] * fq9_pl(i - j);
        r.v[i - 9] = (uint32_t)acc & ZKP_FQ9_MASK;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
// (a * b + c * d) / 2^261 with one reduction (162 + 81 multiply-adds)
ZKP_HD inline fq9 fq9_mul_add2(const fq9& a, const fq9& b, const fq9& c, const fq9& d) {
    uint32_t m[9]; fq9 r; uint64_t acc = 0;
    ZKP_UNROLL for (int i = 0; i < 9; i++) {
        ZKP_UNROLL for (int j = 0; j <= i; j++) { acc += (uint64_t)a.v[j] * b.v[i - j]; acc += (uint64_t)c.v[j] * d.v[i - j]; }
        ZKP_UNROLL for (int j = 0; j < i; j++) acc += (uint64_t)m[j] * fq9_pl(i - j);
        m[i] = ((uint32_t)acc * ZKP_FQ9_N0) & ZKP_FQ9_MASK;
        acc += (uint64_t)m[i] * fq9_pl(0);
        acc >>= 29;
    }
    ZKP_UNROLL for (int i = 9; i < 17; i++) {
        ZKP_UNROLL for (int j = i - 8; j < 9; j++) { acc += (uint64_t)a.v[j] * b.v[i - j]; acc += (uint64_t)c.v[j] * d.v[i - j]; }
        ZKP_UNROLL for (int j = i - 8; j < 9; j++) acc += (uint64_t)m[j] * fq9_pl(i - j);
        r.v[i - 9] = (uint32_t)acc & ZKP_FQ9_MASK;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
// (a b + c d + e f + g h) / 2^261 with one reduction: 36 + 9 products of < 2^58 per column, < 2^63.5 (the Fq2 combination
// R W - Y1 J of the G2 addition is two of these instead of two Fq2 products)
ZKP_HD inline fq9 fq9_mul_add4(const fq9& a, const fq9& b, const fq9& c, const fq9& d, const fq9& e, const fq9& f, const fq9& g, const fq9& h) {
    uint32_t m[9]; fq9 r; uint64_t acc = 0;
    ZKP_UNROLL for (int i = 0; i < 9; i++) {
        ZKP_UNROLL for (int j = 0; j <= i; j++) {
            acc += (uint64_t)a.v[j] * b.v[i - j]; acc += (uint64_t)c.v[j] * d.v[i - j];
            acc += (uint64_t)e.v[j] * f.v[i - j]; acc += (uint64_t)g.v[j] * h.v[i - j];
        }
        ZKP_UNROLL for (int j = 0; j < i; j++) acc += (uint64_t)m[j] * fq9_pl(i - j);
        m[i] = ((uint32_t)acc * ZKP_FQ9_N0) & ZKP_FQ9_MASK;
        acc += (uint64_t)m[i] * fq9_pl(0);
        acc >>= 29;
    }
    ZKP_UNROLL for (int i = 9; i < 17; i++) {
        ZKP_UNROLL for (int j = i - 8; j < 9; j++) {
            acc += (uint64_t)a.v[j] * b.v[i - j]; acc += (uint64_t)c.v[j] * d.v[i - j];
            acc += (uint64_t)e.v[j] * f.v[i - j]; acc += (uint64_t)g.v[j] * h.v[i - j];
        }
        ZKP_UNROLL for (int j = i - 8; j < 9; j++) acc += (uint64_t)m[j] * fq9_pl(i - j);
        r.v[i - 9] = (uint32_t)acc & ZKP_FQ9_MASK;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
// 2 a, limb-wise (limbs < 2^30): allowed as ONE operand of a plain fq9_mul (9 x 2^59 + 9 x 2^58 per column)
ZKP_HD inline fq9 fq9_dbl_l(const fq9& a) { fq9 r; ZKP_UNROLL for (int i = 0; i < 9; i++) r.v[i] = a.v[i] << 1; return r; }
// (neg ? -a : a) - b + K p in one carry pass (a + b < K p)
template <int K> ZKP_HD inline fq9 fq9_sgn_sub_k(bool neg, const fq9& a, const fq9& b) {
    fq9 r; int32_t c = 0;
    ZKP_UNROLL for (int i = 0; i < 8; i++) {
        const int32_t s = neg ? -(int32_t)a.v[i] : (int32_t)a.v[i];
        const int32_t t = s - (int32_t)b.v[i] + (int32_t)fq9_kp<K>(i) + c;
        r.v[i] = (uint32_t)t & ZKP_FQ9_MASK; c = t >> 29;
    }
    const int32_t s8 = neg ? -(int32_t)a.v[8] : (int32_t)a.v[8];
    r.v[8] = (uint32_t)(s8 - (int32_t)b.v[8] + (int32_t)fq9_kp<K>(8) + c);
    return r;
}
// a - b + K p (b < K p), carries propagated: limb differences stay within (-2^29, 2^30), the running carry within {-1, 0, 1}
template <int K> ZKP_HD inline fq9 fq9_sub_k(const fq9& a, const fq9& b) {
    fq9 r; int32_t c = 0;
    ZKP_UNROLL for (int i = 0; i < 8; i++) {
        const int32_t t = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)fq9_kp<K>(i) + c;
        r.v[i] = (uint32_t)t & ZKP_FQ9_MASK; c = t >> 29;
    }
    r.v[8] = (uint32_t)((int32_t)a.v[8] - (int32_t)b.v[8] + (int32_t)fq9_kp<K>(8) + c);
    return r;
}
// a - b - 2 c + K p in one carry pass (b + 2 c < K p): limb sums within (-2^31, 2^30), the running carry within [-4, 1]
template <int K> ZKP_HD inline fq9 fq9_sub2_k(const fq9& a, const fq9& b, const fq9& c2) {
    fq9 r; int32_t c = 0;
    ZKP_UNROLL for (int i = 0; i < 8; i++) {
        const int32_t t = (int32_t)a.v[i] - (int32_t)b.v[i] - (int32_t)(c2.v[i] << 1) + (int32_t)fq9_kp<K>(i) + c;
        r.v[i] = (uint32_t)t & ZKP_FQ9_MASK; c = t >> 29;
    }
    r.v[8] = (uint32_t)((int32_t)a.v[8] - (int32_t)b.v[8] - (int32_t)(c2.v[8] << 1) + (int32_t)fq9_kp<K>(8) + c);
    return r;
}
ZKP_HD inline fq9 fq9_sub2_k4(const fq9& a, const fq9& b, const fq9& c2) { return fq9_sub2_k<4>(a, b, c2); }
// ---- loose differences (round 4).  A column of 9 + 9 products of 29-bit limbs leaves under two bits of the accumulator, so this form has no
// limb-wise lazy sums in general -- but ONE operand of each product of a fused double product may be loose: a - b + K p taken limb by limb
// from a "fat" form of K p (the same integer with every limb below the top in [2^29 - 1, 2^30), so that no limb difference is negative), no
// carry pass.  Limbs of the result: < 2^29 + 2^30 for a - b + K p, < 2^30 for K p - b.  fq9_mul_add2(carried, loose, loose', carried)
// then sums 9 x 2^59.6 + 9 x 2^59 + 9 x 2^58 (reduction) < 2^63.8 per column (tests/test_fq_bounds.py).  Measured on the bare addition
// loop (tools/g1_add_rate.hip loose, profiles/r04_g1_add_rate.jsonl): 16.2 -> 16.8 G additions/s with the two carry passes of Y3 folded away.
template <int K> ZKP_HD constexpr uint32_t fq9_fat(int i) { return i == 0 ? fq9_kp<K>(0) + (1u << 29) : i < 8 ? fq9_kp<K>(i) + (1u << 29) - 1u : fq9_kp<K>(8) - 1u; }
template <int K> ZKP_HD inline fq9 fq9_sub_loose(const fq9& a, const fq9& b) { fq9 r; ZKP_UNROLL for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + fq9_fat<K>(i) - b.v[i]; return r; }
template <int K> ZKP_HD inline fq9 fq9_neg_loose(const fq9& b) { fq9 r; ZKP_UNROLL for (int i = 0; i < 9; i++) r.v[i] = fq9_fat<K>(i) - b.v[i]; return r; }
// K p - a (a < K p)
template <int K> ZKP_HD inline fq9 fq9_neg_k(const fq9& a) { return fq9_sub_k<K>(fq9_zero(), a); }
// a + b, carried (values must leave the sum below 2^261)
ZKP_HD inline fq9 fq9_add(const fq9& a, const fq9& b) {
    fq9 r; uint32_t c = 0;
    ZKP_UNROLL for (int i = 0; i < 8; i++) { const uint32_t t = a.v[i] + b.v[i] + c; r.v[i] = t & ZKP_FQ9_MASK; c = t >> 29; }
    r.v[8] = a.v[8] + b.v[8] + c;
    return r;
}
ZKP_HD inline fq9 fq9_select(bool c, const fq9& a, const fq9& b) { fq9 r; ZKP_UNROLL for (int i = 0; i < 9; i++) r.v[i] = c ? a.v[i] : b.v[i]; return r; }

// ---- conversions.  An fq holds x R10 (R10 = 2^260), an fq9 holds x R9 = 2 x R10: going in is a doubling and a re-slicing of the
// same integer, coming back is one fq9 product with the constant R10 mod p (x R9 * R10 / R9 = x R10) and the re-slicing.
ZKP_HD inline fq9 fq9_reslice(const fq& c) {                    // c carried (limbs 0..8 < 2^26), value < 2^261
    fq9 r;
    ZKP_UNROLL for (int j = 0; j < 9; j++) {
        const int bit = 29 * j, i = bit / 26, sh = bit % 26;
        uint64_t x = (uint64_t)c.v[i] >> sh;
        if (i + 1 < 10) x |= (uint64_t)c.v[i + 1] << (26 - sh);
        if (i + 2 < 10) x |= (uint64_t)c.v[i + 2] << (52 - sh);
        r.v[j] = j < 8 ? ((uint32_t)x & ZKP_FQ9_MASK) : (uint32_t)x;
    }
    return r;
}
ZKP_HD inline fq fq_reslice(const fq9& c) {                     // c carried9
    fq r;
    ZKP_UNROLL for (int j = 0; j < 10; j++) {
        const int bit = 26 * j, i = bit / 29, sh = bit % 29;
        uint64_t x = (uint64_t)c.v[i] >> sh;
        if (i + 1 < 9) x |= (uint64_t)c.v[i + 1] << (29 - sh);
        r.v[j] = j < 9 ? ((uint32_t)x & ZKP_FQ_MASK) : (uint32_t)x;
    }
    return r;
}
ZKP_HD inline fq9 fq9_from_fq(const fq& a) { return fq9_reslice(fq_dbl(a)); }          // value < 3p
// ---- key-table form: a carried9 value below 2^256 (table entries are < 2.4 p < 2^255) as eight 32-bit words -- the same integer,
// re-sliced -- so that a G1 entry (x, y) is 64 bytes and a G2 entry 128: one half / one whole 128-byte line per gather instead of
// the 1.6 lines an 80-byte slot of nine-limb coordinates straddled.  Unpacking is one funnel shift and one mask per limb.
ZKP_HD inline void fq9_pack8(uint32_t w[8], const fq9& a) {
    ZKP_UNROLL for (int k = 0; k < 8; k++) {
        const int bit = 32 * k, i = bit / 29, sh = bit % 29;
        uint64_t x = (uint64_t)a.v[i] >> sh;
        if (i + 1 < 9) x |= (uint64_t)a.v[i + 1] << (29 - sh);
        if (i + 2 < 9) x |= (uint64_t)a.v[i + 2] << (58 - sh);
        w[k] = (uint32_t)x;
    }
}
ZKP_HD inline fq9 fq9_unpack8(const uint32_t w[8]) {
    fq9 r;
    ZKP_UNROLL for (int j = 0; j < 9; j++) {
        const int bit = 29 * j, i = bit / 32, sh = bit % 32;
        uint64_t x = (uint64_t)w[i] >> sh;
        if (i + 1 < 8) x |= (uint64_t)w[i + 1] << (32 - sh);
        r.v[j] = j < 8 ? ((uint32_t)x & ZKP_FQ9_MASK) : (uint32_t)x;
    }
    return r;
}
ZKP_HD inline fq fq9_to_fq(const fq9& a) {                                                   // value < a / 169 + p, then < 3p
    fq9 c; ZKP_UNROLL for (int i = 0; i < 9; i++) c.v[i] = fq9_r10(i);
    return fq_reduce_weak(fq_reslice(fq9_mul(a, c)));
}

}  // namespace zkp
