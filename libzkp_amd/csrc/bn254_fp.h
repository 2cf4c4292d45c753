// BN254 base field Fq and scalar field Fr for gfx950: eight 32-bit words in Montgomery form (R = 2^256), one
// CIOS pass of v_mad_u64_u32 per product.  Both moduli are < 2^254, so values are kept LAZILY reduced in [0, 2p):
// products need no final conditional subtraction, additions/subtractions correct by 2p, and full reduction happens
// only at serialisation / comparison.
// Replaces ark-ff's Fp256<MontBackend> as used by ark-bn254 under /root/reference/src/backend/snark.rs:4-12.
#pragma once
#include "zkp_common.h"

namespace zkp {

struct FqParams {
    static ZKP_HD constexpr uint32_t mod(int i) { constexpr uint32_t m[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u}; return m[i]; }
    static ZKP_HD constexpr uint32_t mod2(int i) { constexpr uint32_t m[8] = {0xb0f9fa8eu, 0x7841182du, 0xd0e3951au, 0x2f02d522u, 0x0302b0bbu, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u}; return m[i]; }
    static ZKP_HD constexpr uint32_t r1(int i) { constexpr uint32_t m[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u}; return m[i]; }
    static ZKP_HD constexpr uint32_t r2(int i) { constexpr uint32_t m[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u}; return m[i]; }
    static ZKP_HD constexpr uint32_t r3(int i) { constexpr uint32_t m[8] = {0xda1530dfu, 0xb1cd6dafu, 0xa7283db6u, 0x62f210e6u, 0x0ada0afbu, 0xef7f0b0cu, 0x2d592544u, 0x20fd6e90u}; return m[i]; }
    static constexpr uint32_t n0 = 0xe4866389u;
};
struct FrParams {
    static ZKP_HD constexpr uint32_t mod(int i) { constexpr uint32_t m[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u}; return m[i]; }
    static ZKP_HD constexpr uint32_t mod2(int i) { constexpr uint32_t m[8] = {0xe0000002u, 0x87c3eb27u, 0xf372e122u, 0x5067d090u, 0x0302b0bau, 0x70a08b6du, 0xc2634053u, 0x60c89ce5u}; return m[i]; }
    static ZKP_HD constexpr uint32_t r1(int i) { constexpr uint32_t m[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u}; return m[i]; }
    static ZKP_HD constexpr uint32_t r2(int i) { constexpr uint32_t m[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u}; return m[i]; }
    static ZKP_HD constexpr uint32_t r3(int i) { constexpr uint32_t m[8] = {0xb4bf0040u, 0x5e94d8e1u, 0x1cfbb6b8u, 0x2a489cbeu, 0xa19fcfedu, 0x893cc664u, 0x7fcc657cu, 0x0cf8594bu}; return m[i]; }
    static constexpr uint32_t n0 = 0xefffffffu;
};

template <class P> struct Fp { uint32_t v[8]; };
using fr = Fp<FrParams>;     // the base field Fq has its own unsaturated representation: bn254_fq.h

template <class P> ZKP_HD inline Fp<P> fp_zero() { Fp<P> r; ZKP_UNROLL for (int i = 0; i < 8; i++) r.v[i] = 0; return r; }
template <class P> ZKP_HD inline Fp<P> fp_one() { Fp<P> r; ZKP_UNROLL for (int i = 0; i < 8; i++) r.v[i] = P::r1(i); return r; }
template <class P> ZKP_HD inline Fp<P> fp_const_r2() { Fp<P> r; ZKP_UNROLL for (int i = 0; i < 8; i++) r.v[i] = P::r2(i); return r; }
template <class P> ZKP_HD inline Fp<P> fp_const_r3() { Fp<P> r; ZKP_UNROLL for (int i = 0; i < 8; i++) r.v[i] = P::r3(i); return r; }

// t - m if t >= m else t, for the 8-word constant m(i)
#define ZKP_FP_COND_SUB(out_, t_, M_)                                                              \
    do {                                                                                           \
        uint32_t d_[8]; uint64_t br_ = 0;                                                          \
        ZKP_UNROLL for (int i_ = 0; i_ < 8; i_++) { uint64_t x_ = (uint64_t)(t_)[i_] - M_(i_) - br_; d_[i_] = (uint32_t)x_; br_ = (x_ >> 32) & 1; } \
        ZKP_UNROLL for (int i_ = 0; i_ < 8; i_++) (out_).v[i_] = br_ ? (t_)[i_] : d_[i_];          \
    } while (0)

// a * b * 2^-256 ; inputs < 2p (or a < 2^256 with b < p), output < 2p
template <class P> ZKP_HD inline Fp<P> fp_mul(const Fp<P>& a, const Fp<P>& b) {
    uint32_t t[9];
    ZKP_UNROLL for (int i = 0; i < 9; i++) t[i] = 0;
    ZKP_UNROLL for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
        ZKP_UNROLL for (int j = 0; j < 8; j++) { c += (uint64_t)a.v[j] * b.v[i] + t[j]; t[j] = (uint32_t)c; c >>= 32; }
        const uint32_t t8 = t[8] + (uint32_t)c;          // no overflow: running value < 2^256 * 2
        const uint32_t m = t[0] * P::n0;
        c = (uint64_t)m * P::mod(0) + t[0]; c >>= 32;
        ZKP_UNROLL for (int j = 1; j < 8; j++) { c += (uint64_t)m * P::mod(j) + t[j]; t[j - 1] = (uint32_t)c; c >>= 32; }
        c += t8; t[7] = (uint32_t)c; t[8] = (uint32_t)(c >> 32);
    }
    Fp<P> r; ZKP_UNROLL for (int i = 0; i < 8; i++) r.v[i] = t[i];   // < 2p < 2^255, t[8] == 0
    return r;
}
template <class P> ZKP_HD inline Fp<P> fp_sq(const Fp<P>& a) { return fp_mul(a, a); }

template <class P> ZKP_HD inline Fp<P> fp_add(const Fp<P>& a, const Fp<P>& b) {   // < 4p -> < 2p
    uint32_t t[8]; uint64_t c = 0;
    ZKP_UNROLL for (int i = 0; i < 8; i++) { c += (uint64_t)a.v[i] + b.v[i]; t[i] = (uint32_t)c; c >>= 32; }
    Fp<P> r; ZKP_FP_COND_SUB(r, t, P::mod2); return r;
}
template <class P> ZKP_HD inline Fp<P> fp_sub(const Fp<P>& a, const Fp<P>& b) {   // a - b (+ 2p if negative) in [0, 2p)
    uint32_t d[8]; uint64_t br = 0;
    ZKP_UNROLL for (int i = 0; i < 8; i++) { uint64_t x = (uint64_t)a.v[i] - b.v[i] - br; d[i] = (uint32_t)x; br = (x >> 32) & 1; }
    uint64_t c = 0; Fp<P> r;
    ZKP_UNROLL for (int i = 0; i < 8; i++) { c += (uint64_t)d[i] + (br ? P::mod2(i) : 0u); r.v[i] = (uint32_t)c; c >>= 32; }
    return r;
}
template <class P> ZKP_HD inline Fp<P> fp_neg(const Fp<P>& a) { return fp_sub(fp_zero<P>(), a); }
template <class P> ZKP_HD inline Fp<P> fp_dbl(const Fp<P>& a) { return fp_add(a, a); }

// full reduction to [0, p)
template <class P> ZKP_HD inline Fp<P> fp_reduce(const Fp<P>& a) { Fp<P> r; ZKP_FP_COND_SUB(r, a.v, P::mod); return r; }
template <class P> ZKP_HD inline bool fp_is_zero(const Fp<P>& a) {
    const Fp<P> r = fp_reduce(a); uint32_t o = 0; ZKP_UNROLL for (int i = 0; i < 8; i++) o |= r.v[i]; return o == 0;
}
template <class P> ZKP_HD inline bool fp_eq(const Fp<P>& a, const Fp<P>& b) { return fp_is_zero(fp_sub(a, b)); }
template <class P> ZKP_HD inline Fp<P> fp_select(bool c, const Fp<P>& a, const Fp<P>& b) {
    Fp<P> r; ZKP_UNROLL for (int i = 0; i < 8; i++) r.v[i] = c ? a.v[i] : b.v[i]; return r;
}

// raw little-endian words <-> Montgomery form
template <class P> ZKP_HD inline Fp<P> fp_from_raw(const uint32_t w[8]) {     // any raw < 2^256 -> Montgomery form of raw mod p
    Fp<P> x; ZKP_UNROLL for (int i = 0; i < 8; i++) x.v[i] = w[i];
    return fp_mul(x, fp_const_r2<P>());
}
template <class P> ZKP_HD inline void fp_to_raw(uint32_t w[8], const Fp<P>& a) {   // canonical value
    Fp<P> one; ZKP_UNROLL for (int i = 0; i < 8; i++) one.v[i] = i == 0 ? 1u : 0u;
    const Fp<P> r = fp_reduce(fp_mul(a, one));
    ZKP_UNROLL for (int i = 0; i < 8; i++) w[i] = r.v[i];
}
template <class P> ZKP_HD inline Fp<P> fp_from_u64(uint64_t x) {
    const uint32_t w[8] = {(uint32_t)x, (uint32_t)(x >> 32), 0, 0, 0, 0, 0, 0};
    return fp_from_raw<P>(w);
}
template <class P> ZKP_HD inline Fp<P> fp_from_wide(const uint32_t w[16]) {   // 512-bit little-endian -> mod p (Montgomery form)
    Fp<P> lo, hi;
    ZKP_UNROLL for (int i = 0; i < 8; i++) { lo.v[i] = w[i]; hi.v[i] = w[8 + i]; }
    return fp_add(fp_mul(lo, fp_const_r2<P>()), fp_mul(hi, fp_const_r3<P>()));
}

// a^(p-2): Fermat ladder (used a handful of times per proof: affine conversion of the three proof points)
template <class P> ZKP_HD inline Fp<P> fp_inv(const Fp<P>& a) {
    Fp<P> acc = fp_one<P>();
    for (int i = 255; i >= 0; i--) {
        acc = fp_sq(acc);
        uint32_t w = P::mod(i >> 5);
        if ((i >> 5) == 0) w -= 2;                 // p - 2 (p is odd and its low word is >= 2)
        if ((w >> (i & 31)) & 1u) acc = fp_mul(acc, a);
    }
    return acc;
}

}  // namespace zkp
