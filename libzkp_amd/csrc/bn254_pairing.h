// BN254 pairing for Groth16 verification (SnarkBackend::verify / verify_membership_zk,
// /root/reference/src/backend/snark.rs:377-401,455-495 -> ark-groth16 verify_with_processed_vk).
// Tower Fq12 = Fq6[w]/(w^2 - v), Fq6 = Fq2[v]/(v^3 - xi), xi = 9 + u (so w^6 = xi, as in oracle/py/bn254.py's
// Fq[w]/(w^12 - 18 w^6 + 82)).  The pairing is the plain ate pairing a(Q, P) = f_{t-1,Q}(P)^((p^12-1)/r) with
// t - 1 = 6 x^2: inversion-free Miller loop in Jacobian coordinates with denominator elimination; final exponentiation =
// easy part (one Fq12 inversion, conjugation, the p^2 Frobenius) and a 761-bit square-and-multiply for the hard part.  It is a different
// bilinear map from the optimal ate pairing ark-ec uses (a fixed power of it), which does not matter for a verifier:
// the product-of-pairings check e(A,B) e(-alpha,beta) e(-L,gamma) e(-C,delta) == 1 holds under one iff under the other.
// Written for clarity, not speed (always-reduced Fq2 operations): verification is not on the proving hot path.
#pragma once
#include "bn254_g.h"

namespace zkp {

struct fq6 { fq2 a0, a1, a2; };
struct fq12 { fq6 c0, c1; };

ZKP_HD inline fq2 fq2_zero() { fq2 r; f_set_zero(r); return r; }
ZKP_HD inline fq2 fq2_one() { fq2 r; f_set_one(r); return r; }
ZKP_HD inline fq2 fq2_mul_fq(const fq2& a, const fq& k) { return fq2{fq_reduce_weak(fq_mul(a.c0, k)), fq_reduce_weak(fq_mul(a.c1, k))}; }
ZKP_HD inline fq2 fq2_mul_xi(const fq2& a) {           // (a0 + a1 u)(9 + u) = (9 a0 - a1) + (a0 + 9 a1) u
    const fq2 a2 = f_dbl(a), a4 = f_dbl(a2), a8 = f_dbl(a4), a9 = f_add(a8, a);
    return fq2{fq_sub(a9.c0, a.c1), fq_add(a9.c1, a.c0)};
}
ZKP_HD inline bool fq2_eq(const fq2& a, const fq2& b) { return fq_eq(a.c0, b.c0) && fq_eq(a.c1, b.c1); }

ZKP_HD inline fq6 fq6_zero() { return fq6{fq2_zero(), fq2_zero(), fq2_zero()}; }
ZKP_HD inline fq6 fq6_one() { return fq6{fq2_one(), fq2_zero(), fq2_zero()}; }
ZKP_HD inline fq6 fq6_add(const fq6& a, const fq6& b) { return fq6{f_add(a.a0, b.a0), f_add(a.a1, b.a1), f_add(a.a2, b.a2)}; }
ZKP_HD inline fq6 fq6_sub(const fq6& a, const fq6& b) { return fq6{f_sub(a.a0, b.a0), f_sub(a.a1, b.a1), f_sub(a.a2, b.a2)}; }
ZKP_HD inline fq6 fq6_mul_v(const fq6& a) { return fq6{fq2_mul_xi(a.a2), a.a0, a.a1}; }
ZKP_HD_NOINLINE inline fq6 fq6_mul(const fq6& a, const fq6& b) {
    const fq2 v0 = f_mul(a.a0, b.a0), v1 = f_mul(a.a1, b.a1), v2 = f_mul(a.a2, b.a2);
    const fq2 t0 = f_sub(f_sub(f_mul(f_add(a.a1, a.a2), f_add(b.a1, b.a2)), v1), v2);
    const fq2 t1 = f_sub(f_sub(f_mul(f_add(a.a0, a.a1), f_add(b.a0, b.a1)), v0), v1);
    const fq2 t2 = f_sub(f_sub(f_mul(f_add(a.a0, a.a2), f_add(b.a0, b.a2)), v0), v2);
    return fq6{f_add(v0, fq2_mul_xi(t0)), f_add(t1, fq2_mul_xi(v2)), f_add(t2, v1)};
}
ZKP_HD inline fq12 fq12_one() { return fq12{fq6_one(), fq6_zero()}; }
ZKP_HD_NOINLINE inline fq12 fq12_mul(const fq12& a, const fq12& b) {
    const fq6 t0 = fq6_mul(a.c0, b.c0), t1 = fq6_mul(a.c1, b.c1);
    const fq6 m = fq6_mul(fq6_add(a.c0, a.c1), fq6_add(b.c0, b.c1));
    return fq12{fq6_add(t0, fq6_mul_v(t1)), fq6_sub(fq6_sub(m, t0), t1)};
}
ZKP_HD_NOINLINE inline fq12 fq12_sq(const fq12& a) {      // (c0 + c1 w)^2 = (c0 + c1)(c0 + v c1) - c0c1 - v c0c1 + 2 c0c1 w
    const fq6 t = fq6_mul(a.c0, a.c1);
    const fq6 s = fq6_mul(fq6_add(a.c0, a.c1), fq6_add(a.c0, fq6_mul_v(a.c1)));
    return fq12{fq6_sub(fq6_sub(s, t), fq6_mul_v(t)), fq6_add(t, t)};
}
ZKP_HD inline bool fq12_is_one(const fq12& a) {
    return fq2_eq(a.c0.a0, fq2_one()) && f_is_zero(a.c0.a1) && f_is_zero(a.c0.a2) && f_is_zero(a.c1.a0) && f_is_zero(a.c1.a1) && f_is_zero(a.c1.a2);
}
// the line l = A + (B + C v) w  (A, B, C in Fq2): three of the six Fq2 coefficients of an Fq12 element
struct fq12_line { fq2 A, B, C; };
ZKP_HD inline fq12 fq12_from_line(const fq2& A, const fq2& B, const fq2& C) { return fq12{fq6{A, fq2_zero(), fq2_zero()}, fq6{B, C, fq2_zero()}}; }
ZKP_HD inline fq6 fq6_mul_fq2(const fq6& a, const fq2& k) { return fq6{f_mul(a.a0, k), f_mul(a.a1, k), f_mul(a.a2, k)}; }
// a * (b0 + b1 v): five Fq2 products
ZKP_HD inline fq6 fq6_mul_sparse2(const fq6& a, const fq2& b0, const fq2& b1) {
    const fq2 v0 = f_mul(a.a0, b0), v1 = f_mul(a.a1, b1);
    const fq2 mid = f_sub(f_sub(f_mul(f_add(a.a0, a.a1), f_add(b0, b1)), v0), v1);      // a0 b1 + a1 b0
    return fq6{f_add(v0, fq2_mul_xi(f_mul(a.a2, b1))), mid, f_add(v1, f_mul(a.a2, b0))};
}
// f * l for a line: 13 Fq2 products instead of the 18 of a general product
ZKP_HD_NOINLINE inline fq12 fq12_mul_line(const fq12& f, const fq12_line& l) {
    const fq6 t0 = fq6_mul_fq2(f.c0, l.A), t1 = fq6_mul_sparse2(f.c1, l.B, l.C);
    const fq6 m = fq6_mul_sparse2(fq6_add(f.c0, f.c1), f_add(l.A, l.B), l.C);
    return fq12{fq6_add(t0, fq6_mul_v(t1)), fq6_sub(fq6_sub(m, t0), t1)};
}

// tangent at T (Jacobian over Fq2, twisted curve) evaluated at P = (xp, yp) in G1, scaled by subfield factors:
//   2 Y Z^3 yp  -  3 X^2 Z^2 xp w  +  (3 X^3 - 2 Y^2) w^3        (w^3 = v w)
ZKP_HD_NOINLINE inline fq12_line line_double(const g2_jac& T, const fq& xp, const fq& yp) {
    const fq2 XX = f_sq(T.X), YY = f_sq(T.Y), ZZ = f_sq(T.Z);
    const fq2 A = fq2_mul_fq(f_dbl(f_mul(f_mul(T.Y, T.Z), ZZ)), yp);
    const fq2 x3 = f_add(f_dbl(XX), XX);                                  // 3 X^2
    const fq2 B = f_neg(fq2_mul_fq(f_mul(x3, ZZ), xp));
    const fq2 C = f_sub(f_mul(x3, T.X), f_dbl(YY));
    return fq12_line{A, B, C};
}
// chord through T (Jacobian) and Q (affine):  D yp - N xp w + (N x2 - D y2) w^3,  N = y2 Z^3 - Y,  D = (x2 Z^2 - X) Z
ZKP_HD_NOINLINE inline fq12_line line_add(const g2_jac& T, const g2_aff& Q, const fq& xp, const fq& yp) {
    const fq2 ZZ = f_sq(T.Z);
    const fq2 N = f_sub(f_mul(f_mul(Q.y, T.Z), ZZ), T.Y);
    const fq2 D = f_mul(f_sub(f_mul(Q.x, ZZ), T.X), T.Z);
    return fq12_line{fq2_mul_fq(D, yp), f_neg(fq2_mul_fq(N, xp)), f_sub(f_mul(N, Q.x), f_mul(D, Q.y))};
}
// f_{t-1,Q}(P), t - 1 = 6 x^2 = 0x6f4d8248eeb859fbf83e9682e87cfd46 (127 bits)
ZKP_HD_NOINLINE inline fq12 miller_loop(const g2_aff& Q, const g1_aff& P) {
    const uint32_t T1[4] = {0xe87cfd46u, 0xf83e9682u, 0xeeb859fbu, 0x6f4d8248u};
    fq12 f = fq12_one();
    g2_jac T = jac_from_aff(Q);
    for (int i = 125; i >= 0; i--) {
        f = fq12_mul_line(fq12_sq(f), line_double(T, P.x, P.y));
        T = jac_dbl(T);
        if ((T1[i >> 5] >> (i & 31)) & 1u) {
            f = fq12_mul_line(f, line_add(T, Q, P.x, P.y));
            T = jac_madd(T, Q);
        }
    }
    return f;
}
ZKP_HD constexpr uint32_t final_exp_word(int i) {
    constexpr uint32_t E[88] = {
        0xca86f120u, 0x86964b64u, 0xe54523a4u, 0x40a4efb7u, 0x96e84abbu, 0x837fa978u, 0xb9b2b918u, 0x361102b6u,
        0xf35692dau, 0xc0de81deu, 0xa6c3c760u, 0xbe04c7e8u, 0xd570bb7fu, 0xd766f9c9u, 0x83561841u, 0xc230974du,
        0xc3be69a3u, 0x5bba1668u, 0x10526294u, 0x7f3811c4u, 0xdadda71cu, 0x29baee7du, 0x145da900u, 0xbf813b8du,
        0x423f9a2cu, 0x641bbadfu, 0x44eacc5eu, 0xa80bb4eau, 0x14fde37cu, 0xcd656648u, 0x580291d2u, 0x4a0364b9u,
        0x0826f0ddu, 0xee93dfb1u, 0xc5514724u, 0x6b42db8du, 0x0b0f3785u, 0xbb10cf43u, 0x6f804216u, 0x40494e40u,
        0xacf3aafbu, 0x55cfe107u, 0xe0ebae87u, 0x2088ec80u, 0x11a337a0u, 0x846a3ed0u, 0x1e3a5195u, 0x48a45a4au,
        0xdfc50e16u, 0xe5664568u, 0x4c0cc4ebu, 0xab6a4129u, 0xd268c7dau, 0x82d0d602u, 0xed3cc48au, 0x6668449au,
        0xb2015dfcu, 0x5062cd0fu, 0xb1ddb3d1u, 0x7f2940a8u, 0x2a226448u, 0x77f5b63au, 0x61e443aeu, 0xfef07813u,
        0x88d5c6c8u, 0xf977870eu, 0x1f676baau, 0x790364a6u, 0xceaddea3u, 0x5887e72eu, 0xa09a1b70u, 0x1377e563u,
        0x1bd8c3b2u, 0x0c54efeeu, 0xd524d8f7u, 0x3ec3d15au, 0xb2383a5du, 0xdaf15466u, 0xbb94fec0u, 0xe1e30a73u,
        0x5f3f7be2u, 0x6a1c7101u, 0x6369b1ffu, 0x842d43bfu, 0x107d20bcu, 0x20fddadfu, 0x4b6dc970u, 0x0000002fu
    };
    return E[i];
}
// ---- final exponentiation f^((p^12 - 1)/r) = ((f^(p^6 - 1))^(p^2 + 1))^((p^4 - p^2 + 1)/r)
ZKP_HD inline fq6 fq6_neg(const fq6& a) { return fq6{f_neg(a.a0), f_neg(a.a1), f_neg(a.a2)}; }
ZKP_HD_NOINLINE inline fq6 fq6_inv(const fq6& a) {
    const fq2 A = f_sub(f_sq(a.a0), fq2_mul_xi(f_mul(a.a1, a.a2)));
    const fq2 B = f_sub(fq2_mul_xi(f_sq(a.a2)), f_mul(a.a0, a.a1));
    const fq2 C = f_sub(f_sq(a.a1), f_mul(a.a0, a.a2));
    const fq2 F = f_add(f_mul(a.a0, A), fq2_mul_xi(f_add(f_mul(a.a2, B), f_mul(a.a1, C))));
    return fq6_mul_fq2(fq6{A, B, C}, f_inv(F));
}
ZKP_HD_NOINLINE inline fq12 fq12_inv(const fq12& a) {        // (c0 - c1 w) / (c0^2 - v c1^2)
    const fq6 t = fq6_inv(fq6_sub(fq6_mul(a.c0, a.c0), fq6_mul_v(fq6_mul(a.c1, a.c1))));
    return fq12{fq6_mul(a.c0, t), fq6_neg(fq6_mul(a.c1, t))};
}
ZKP_HD inline fq12 fq12_conj(const fq12& a) { return fq12{a.c0, fq6_neg(a.c1)}; }     // the p^6-power Frobenius: w -> -w
// the p^2-power Frobenius fixes Fq2 and maps w^k to gamma_k w^k with gamma_k = xi^((p^2 - 1) k / 6) in Fq
ZKP_HD_NOINLINE inline fq12 fq12_frob_p2(const fq12& a) {
    const uint32_t G[5][8] = {
        {0x607cfd49u, 0xe4bd44e5u, 0xbb966e3du, 0xc28f069fu, 0xe0acccb0u, 0x5e6dd9e7u, 0xe131a029u, 0x30644e72u},
        {0x607cfd48u, 0xe4bd44e5u, 0xbb966e3du, 0xc28f069fu, 0xe0acccb0u, 0x5e6dd9e7u, 0xe131a029u, 0x30644e72u},
        {0xd87cfd46u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u},
        {0x77fffffeu, 0x57634731u, 0xacdb5c4fu, 0xd4f263f1u, 0xa0d48bacu, 0x59e26bceu, 0x00000000u, 0x00000000u},
        {0x77ffffffu, 0x57634731u, 0xacdb5c4fu, 0xd4f263f1u, 0xa0d48bacu, 0x59e26bceu, 0x00000000u, 0x00000000u}};
    fq g[5]; for (int k = 0; k < 5; k++) g[k] = fq_from_raw(G[k]);
    return fq12{fq6{a.c0.a0, fq2_mul_fq(a.c0.a1, g[1]), fq2_mul_fq(a.c0.a2, g[3])}, fq6{fq2_mul_fq(a.c1.a0, g[0]), fq2_mul_fq(a.c1.a1, g[2]), fq2_mul_fq(a.c1.a2, g[4])}};
}
ZKP_HD constexpr uint32_t hard_exp_word(int i) {
    constexpr uint32_t E[24] = {
        0xccdf42b1u, 0xe81bb482u, 0xf49c36d4u, 0x5abf5cc4u, 0x1da014fdu, 0xf1154e7eu, 0x87cdbacfu, 0xdcc7b44cu,
        0x954bcf8au, 0xaaa441e3u, 0xd5095f23u, 0x6b887d56u, 0xf3fd90c6u, 0x79581e16u, 0xd189227du, 0x3b1b1355u,
        0x61876f6bu, 0x4e529a58u, 0xd5b12278u, 0x6c0eb522u, 0x83177fafu, 0x331ec151u, 0x0b0759adu, 0x01baaa71u
    };
    return E[i];
}
ZKP_HD_NOINLINE inline fq12 final_exponentiation(const fq12& f) {
    const fq12 f1 = fq12_mul(fq12_conj(f), fq12_inv(f));               // f^(p^6 - 1)
    const fq12 f2 = fq12_mul(fq12_frob_p2(f1), f1);                   // ^(p^2 + 1)
    fq12 acc = f2;                                                   // ^((p^4 - p^2 + 1)/r): 761-bit square-and-multiply, top bit set
    for (int i = 759; i >= 0; i--) {
        acc = fq12_sq(acc);
        if ((hard_exp_word(i >> 5) >> (i & 31)) & 1u) acc = fq12_mul(acc, f2);
    }
    return acc;
}
// ---- the same check with the hard part as an addition chain in x (BN parameter, p = 36x^4 + 36x^3 + 24x^2 + 6x + 1):
// f -> f^(m (p^12 - 1)/r) with m = 2x(6x^2 + 3x + 1), which is 1 exactly when f^((p^12 - 1)/r) is (m is prime to r).  Three
// 63-bit powers and the p, p^2, p^3 Frobenius maps replace the 761-bit power: ~190 squarings and ~100 products.
// The p^j-power Frobenius conjugates the Fq2 coefficients (odd j) and maps w^k to gamma_{j,k} w^k.
ZKP_HD inline fq2 fq2_conj(const fq2& a) { return fq2{a.c0, fq_neg(a.c1)}; }
ZKP_HD_NOINLINE inline fq12 fq12_frob_odd(const fq12& a, int j) {      // j = 1 or 3
    // gamma_{1,k} = xi^(k (p^1 - 1) / 6), k = 1..5: (c0, c1) raw words
    static constexpr uint32_t G1[5][2][8] = {
        {{0xdcc9e470u, 0xd60b35dau, 0x292f2176u, 0x5c521e08u, 0x76e68b60u, 0xe8b99fddu, 0x2865a7dfu, 0x1284b71cu},
         {0x80f362acu, 0xca5cf05fu, 0x8eeec7e5u, 0x74799277u, 0x12150b8eu, 0xa6327cfeu, 0xb4fae7e6u, 0x246996f3u}},
        {{0x176f553du, 0x99e39557u, 0xc2c3330cu, 0xb78cc310u, 0xf559b143u, 0x4c0bec3cu, 0x4f7911f7u, 0x2fb34798u},
         {0x640fcba2u, 0x1665d51cu, 0x0b7c9dceu, 0x32ae2a1du, 0xd75a0794u, 0x4ba4cc8bu, 0x61ebae20u, 0x16c9e550u}},
        {{0x71a0135au, 0xdc540146u, 0xa9c95998u, 0xdbaae0edu, 0xb6e2f9b9u, 0xdc5ec698u, 0x489af5dcu, 0x063cf305u},
         {0x2623b0e3u, 0x82d37f63u, 0x8fa25bd2u, 0x21807dc9u, 0xec796f2bu, 0x0704b5a7u, 0xac41049au, 0x07c03cbcu}},
        {{0x921ea762u, 0x848a1f55u, 0xbe94ec72u, 0xd33365f7u, 0x5a181e84u, 0x80f3c0b7u, 0x64eea801u, 0x05b54f5eu},
         {0xcd2b8126u, 0xc13b4711u, 0x1bdec763u, 0x3685d2eau, 0x3b0b1c92u, 0x9f3a80b0u, 0xe7fd8aeeu, 0x2c145edbu}},
        {{0xeab7692fu, 0x2ea2c810u, 0x55aa1bd3u, 0x425c459bu, 0xa4353ff4u, 0xe93a3661u, 0x4f798649u, 0x0183c1e7u},
         {0x6e0c2c4bu, 0x24c6b8eeu, 0x678e2ac0u, 0xb080cb99u, 0xc7729f7du, 0xa27fb246u, 0x76fd0675u, 0x12acf2cau}}
    };
    // gamma_{3,k} = xi^(k (p^3 - 1) / 6), k = 1..5: (c0, c1) raw words
    static constexpr uint32_t G3[5][2][8] = {
        {{0x1ed4a67fu, 0xe86f7d39u, 0xbe55d24au, 0x894cb38du, 0xd0acaa90u, 0xefe9608cu, 0xcc82e4bbu, 0x19dc81cfu},
         {0xf4c0c101u, 0x7694aa2bu, 0x97d439ecu, 0x7f03a5e3u, 0x3576139du, 0x06cbeee3u, 0x0be77d73u, 0x00abf8b6u}},
        {{0x7bdcfb6du, 0x7b746ee8u, 0x5d6942d3u, 0x805ffd3du, 0x959f25acu, 0xbaff1c77u, 0xb755ef0au, 0x0856e078u},
         {0xaaa586deu, 0x380cab2bu, 0x98ff2631u, 0x0fdf31bfu, 0xec26094fu, 0xa9f30e6du, 0xb3d1766fu, 0x04f1de41u}},
        {{0x66dce9edu, 0x5fcc8ad0u, 0xbea870f4u, 0xbbd689a3u, 0xca9e5ea3u, 0xdbf17f1du, 0x9896aa4cu, 0x2a275b6du},
         {0xb2594c64u, 0xb94d0cb3u, 0xd8cf6ebau, 0x7600ecc7u, 0x9507e932u, 0xb14b900eu, 0x34f09b8fu, 0x28a411b6u}},
        {{0x3ccbf066u, 0x0e1a92bcu, 0x75b06bcbu, 0xe6330945u, 0xb5b2444eu, 0x19bee0f7u, 0x11c08dabu, 0x0bc58c66u},
         {0x730c239fu, 0x5fe3ed9du, 0x737f96e5u, 0xa44a9e08u, 0x0cd21d04u, 0xfeb0f6efu, 0xe1910a12u, 0x23d5e999u}},
        {{0x76261b43u, 0xebde8470u, 0x967c84a5u, 0x2ed68098u, 0x3b4d3f69u, 0x711699fau, 0x952c0905u, 0x13c49044u},
         {0x84282499u, 0x1f250413u, 0x20028021u, 0x3e2ddaeau, 0x2a48633du, 0x9fb1b228u, 0x59b1dd0bu, 0x16db366au}}
    };
    fq2 g[5];
    for (int k = 0; k < 5; k++) { const auto& G = j == 1 ? G1[k] : G3[k]; g[k] = fq2{fq_from_raw(G[0]), fq_from_raw(G[1])}; }
    return fq12{fq6{fq2_conj(a.c0.a0), f_mul(fq2_conj(a.c0.a1), g[1]), f_mul(fq2_conj(a.c0.a2), g[3])},
                fq6{f_mul(fq2_conj(a.c1.a0), g[0]), f_mul(fq2_conj(a.c1.a1), g[2]), f_mul(fq2_conj(a.c1.a2), g[4])}};
}
// Squaring inside the cyclotomic subgroup (Granger-Scott): with s = w^3, t = w the element is A + B t + C t^2 over
// Fq4 = Fq2[s]/(s^2 - xi), A = a0 + b1 s, B = b0 + a2 s, C = a1 + b2 s (a_i, b_i the Fq2 coefficients of c0, c1), and
// f^2 = (3A^2 - 2 conj A) + (3 s C^2 + 2 conj B) t + (3B^2 - 2 conj C) t^2: three Fq4 squarings = 9 Fq2 squarings, half the
// work of the general fq12_sq.  Only valid after the easy part of the final exponentiation.
ZKP_HD inline void fq4_sq(fq2& re, fq2& im, const fq2& x, const fq2& y) {       // (x + y s)^2 = (x^2 + xi y^2) + 2xy s
    const fq2 xx = f_sq(x), yy = f_sq(y);
    re = f_add(xx, fq2_mul_xi(yy));
    im = f_sub(f_sub(f_sq(f_add(x, y)), xx), yy);
}
ZKP_HD_NOINLINE inline fq12 fq12_cyclo_sq(const fq12& f) {
    fq2 t0, t1, t2, t3, t4, t5;
    fq4_sq(t0, t1, f.c0.a0, f.c1.a1);
    fq4_sq(t2, t3, f.c1.a0, f.c0.a2);
    fq4_sq(t4, t5, f.c0.a1, f.c1.a2);
    auto three_minus_two = [](const fq2& t, const fq2& z) { return f_sub(f_add(f_dbl(t), t), f_dbl(z)); };   // 3t - 2z
    auto three_plus_two = [](const fq2& t, const fq2& z) { return f_add(f_add(f_dbl(t), t), f_dbl(z)); };    // 3t + 2z
    fq12 r;
    r.c0.a0 = three_minus_two(t0, f.c0.a0); r.c1.a1 = three_plus_two(t1, f.c1.a1);
    r.c1.a0 = three_plus_two(fq2_mul_xi(t5), f.c1.a0); r.c0.a2 = three_minus_two(t4, f.c0.a2);
    r.c0.a1 = three_minus_two(t2, f.c0.a1); r.c1.a2 = three_plus_two(t3, f.c1.a2);
    return r;
}
ZKP_HD_NOINLINE inline fq12 fq12_pow_x(const fq12& a) {                // a^x, x = 4965661367192848881 (63 bits); a cyclotomic
    const uint64_t X = 4965661367192848881ull;
    fq12 acc = a;
    for (int i = 61; i >= 0; i--) {
        acc = fq12_cyclo_sq(acc);
        if ((X >> i) & 1u) acc = fq12_mul(acc, a);
    }
    return acc;
}
// (inside the cyclotomic subgroup the inverse is the conjugate)
ZKP_HD_NOINLINE inline fq12 final_exponentiation_chain(const fq12& f) {
    const fq12 e1 = fq12_mul(fq12_conj(f), fq12_inv(f));               // f^(p^6 - 1)
    const fq12 r = fq12_mul(fq12_frob_p2(e1), e1);                    // ^(p^2 + 1)
    const fq12 y0 = fq12_conj(fq12_pow_x(r));                          // r^-x
    const fq12 y1 = fq12_cyclo_sq(y0);
    const fq12 y3 = fq12_mul(fq12_cyclo_sq(y1), y1);
    const fq12 y4 = fq12_conj(fq12_pow_x(y3));
    const fq12 y6 = fq12_pow_x(fq12_cyclo_sq(y4));                     // (y5^-x)^-1
    const fq12 y8 = fq12_mul(fq12_mul(y6, y4), fq12_conj(y3));
    const fq12 y9 = fq12_mul(y8, y1);
    const fq12 y11 = fq12_mul(fq12_mul(y8, y4), r);
    const fq12 y13 = fq12_mul(fq12_frob_odd(y9, 1), y11);
    const fq12 y14 = fq12_mul(fq12_frob_p2(y8), y13);
    return fq12_mul(fq12_frob_odd(fq12_mul(fq12_conj(r), y9), 3), y14);
}
// the same power as one 2790-bit square-and-multiply (no inversion, no Frobenius): the test suite's cross-check
ZKP_HD_NOINLINE inline fq12 final_exponentiation_naive(const fq12& f) {
    fq12 acc = fq12_one();
    bool started = false;
    for (int i = 2789; i >= 0; i--) {
        if (started) acc = fq12_sq(acc);
        if ((final_exp_word(i >> 5) >> (i & 31)) & 1u) { acc = started ? fq12_mul(acc, f) : f; started = true; }
    }
    return acc;
}

}  // namespace zkp
