// BN254 scalar field Fr in nine 29-bit limbs (Montgomery form, R9 = 2^261) for the QAP step of the Groth16 prover (the seven
// size-m transforms per proof in LDS): a product is 81 + 81 multiply-adds with one 64-bit accumulator per column, where the
// saturated 8 x 32 CIOS form of bn254_fp.h spends ~600 instructions on carry bookkeeping around its 128.  Same rules as
// bn254_fq9.h: operands are CARRIED (every limb < 2^29), values may run far above r (2^261 = 169.3 r), additions and
// subtractions propagate carries.  Witness generation, the CSR products and the digit recoding stay in bn254_fp.h's form;
// fr9_from_fr / fr9_to_fr convert (x R9 = 32 x R256: a 5-bit shift while re-slicing, then one product with R9 mod r to bring
// the value down; back: one product with 2^256 mod r).
//
// Bounds (tests/test_fq_bounds.py): fr9_mul: value < a b / (169.28 r) + r.  A butterfly of a decimation-in-time stage adds a
// product (< 2 r for operands below 84 r) to an unreduced element, so elements grow by <= 2 r per stage (fr9_sub_k<2>); the
// decimation-in-frequency stages add two unreduced elements, which doubles the bound per stage, so their sums go through
// fr9_reduce_weak (< 2.3 r for any input below 2^261).
#pragma once
#include "bn254_fp.h"

namespace zkp {

struct fr9 { uint32_t v[9]; };
#define ZKP_FR9_MASK 0x1fffffffu
#define ZKP_FR9_N0 0xfffffffu          // -r^-1 mod 2^29
#define ZKP_FR9_RECIP 1354u            // floor(2^264 / r)
ZKP_HD constexpr uint32_t fr9_pl(int i) { constexpr uint32_t m[9] = {0x10000001u, 0x1f0fac9fu, 0xe5c2450u, 0x7d090f3u, 0x1585d283u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu}; return m[i]; }
ZKP_HD constexpr uint32_t fr9_k2(int i) { constexpr uint32_t m[9] = {0x2u, 0x1e1f593fu, 0x1cb848a1u, 0xfa121e6u, 0xb0ba506u, 0x5b68181u, 0x14dc282u, 0x1cb84c68u, 0x60c89cu}; return m[i]; }
ZKP_HD constexpr uint32_t fr9_k4(int i) { constexpr uint32_t m[9] = {0x4u, 0x1c3eb27eu, 0x19709143u, 0x1f4243cdu, 0x16174a0cu, 0xb6d0302u, 0x29b8504u, 0x197098d0u, 0xc19139u}; return m[i]; }
ZKP_HD constexpr uint32_t fr9_k32(int i) { constexpr uint32_t m[9] = {0x20u, 0x1f593f0u, 0xb848a1fu, 0x1a121e6eu, 0x10ba5067u, 0x1b681815u, 0x14dc2822u, 0xb84c680u, 0x60c89ceu}; return m[i]; }
ZKP_HD constexpr uint32_t fr9_one(int i) { constexpr uint32_t m[9] = {0xfffff57u, 0x1ea70ab4u, 0x52c068bu, 0x17504f49u, 0xaa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x52ac7a8u, 0xdc836u}; return m[i]; }     // 2^261 mod r
ZKP_HD constexpr uint32_t fr9_r256(int i) { constexpr uint32_t m[9] = {0xffffffbu, 0x4b1a0e2u, 0x18334a6bu, 0x18ed2b3eu, 0x1462e36fu, 0x11b7bc3cu, 0x1cbd99bau, 0x183340fbu, 0xe0a77u}; return m[i]; }    // 2^256 mod r
template <int K> ZKP_HD constexpr uint32_t fr9_kp(int i) {
    static_assert(K == 2 || K == 4 || K == 32, "multiples of r held as constants");
    return K == 2 ? fr9_k2(i) : K == 4 ? fr9_k4(i) : fr9_k32(i);
}

// a * b / 2^261 mod r
ZKP_HD inline fr9 fr9_mul(const fr9& a, const fr9& b) {
    uint32_t m[9]; fr9 r; uint64_t acc = 0;
    ZKP_UNROLL for (int i = 0; i < 9; i++) {
        ZKP_UNROLL for (int j = 0; j <= i; j++) acc += (uint64_t)a.v[j] * b.v[i - j];
        ZKP_UNROLL for (int j = 0; j < i; j++) acc += (uint64_t)m[j] * fr9_pl(i - j);
        m[i] = ((uint32_t)acc * ZKP_FR9_N0) & ZKP_FR9_MASK;
        acc += (uint64_t)m[i] * fr9_pl(0);
        acc >>= 29;
    }
    ZKP_UNROLL for (int i = 9; i < 17; i++) {
        ZKP_UNROLL for (int j = i - 8; j < 9; j++) acc += (uint64_t)a.v[j] * b.v[i - j];
        ZKP_UNROLL for (int j = i - 8; j < 9; j++) acc += (uint64_t)m[j] * fr9_pl(i - j);
        r.v[i - 9] = (uint32_t)acc & ZKP_FR9_MASK;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
// a + b, carried (the sum must stay below 2^261)
ZKP_HD inline fr9 fr9_add(const fr9& a, const fr9& b) {
    fr9 r; uint32_t c = 0;
    ZKP_UNROLL for (int i = 0; i < 8; i++) { const uint32_t t = a.v[i] + b.v[i] + c; r.v[i] = t & ZKP_FR9_MASK; c = t >> 29; }
    r.v[8] = a.v[8] + b.v[8] + c;
    return r;
}
// a - b + K r (b < K r), carried
template <int K> ZKP_HD inline fr9 fr9_sub_k(const fr9& a, const fr9& b) {
    fr9 r; int32_t c = 0;
    ZKP_UNROLL for (int i = 0; i < 8; i++) {
        const int32_t t = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)fr9_kp<K>(i) + c;
        r.v[i] = (uint32_t)t & ZKP_FR9_MASK; c = t >> 29;
    }
    r.v[8] = (uint32_t)((int32_t)a.v[8] - (int32_t)b.v[8] + (int32_t)fr9_kp<K>(8) + c);
    return r;
}
// any carried value < 2^261 -> [0, 2.3 r): subtract q r with q = floor(top limb * floor(2^264 / r) / 2^32) <= floor(a / r)
ZKP_HD inline fr9 fr9_reduce_weak(const fr9& a) {
    const uint32_t q = (uint32_t)(((uint64_t)a.v[8] * ZKP_FR9_RECIP) >> 32);
    fr9 r; int64_t acc = 0;
    ZKP_UNROLL for (int i = 0; i < 8; i++) {
        acc += (int64_t)a.v[i] - (int64_t)((uint64_t)q * fr9_pl(i));
        r.v[i] = (uint32_t)acc & ZKP_FR9_MASK; acc >>= 29;
    }
    r.v[8] = (uint32_t)(acc + (int64_t)a.v[8] - (int64_t)((uint64_t)q * fr9_pl(8)));
    return r;
}

// ---- conversions with bn254_fp.h's eight 32-bit words (Montgomery form with R = 2^256, value < 2r)
template <class P> ZKP_HD inline fr9 fr9_from_fr(const Fp<P>& a) {
    // the integer 32 * a, re-sliced: bit b of a lands at bit b + 5
    fr9 s;
    ZKP_UNROLL for (int j = 0; j < 9; j++) {
        const int bit = 29 * j - 5;                              // first bit of `a` in limb j (negative: the low 5 bits are zero)
        uint64_t x;
        if (bit < 0) x = (uint64_t)a.v[0] << 5;
        else {
            const int wd = bit >> 5, sh = bit & 31;
            x = (uint64_t)a.v[wd] >> sh;
            if (wd + 1 < 8) x |= (uint64_t)a.v[wd + 1] << (32 - sh);
        }
        s.v[j] = j < 8 ? ((uint32_t)x & ZKP_FR9_MASK) : (uint32_t)x;
    }
    fr9 one; ZKP_UNROLL for (int i = 0; i < 9; i++) one.v[i] = fr9_one(i);
    return fr9_mul(s, one);                                      // the same residue, value < 64 r / 169 + r
}
template <class P> ZKP_HD inline Fp<P> fr9_to_fr(const fr9& a) {  // value < a / 169 + r < 2r for a < 169 r
    fr9 c; ZKP_UNROLL for (int i = 0; i < 9; i++) c.v[i] = fr9_r256(i);
    const fr9 t = fr9_mul(a, c);
    Fp<P> r;
    ZKP_UNROLL for (int wd = 0; wd < 8; wd++) {
        const int bit = 32 * wd, i = bit / 29, sh = bit % 29;
        uint64_t x = (uint64_t)t.v[i] >> sh;
        if (i + 1 < 9) x |= (uint64_t)t.v[i + 1] << (29 - sh);
        if (i + 2 < 9) x |= (uint64_t)t.v[i + 2] << (58 - sh);
        r.v[wd] = (uint32_t)x;
    }
    return r;
}

}  // namespace zkp
