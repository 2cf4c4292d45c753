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
// the line l = A + (B + C v) w  (A, B, C in Fq2) as an Fq12 element
ZKP_HD inline fq12 fq12_from_line(const fq2& A, const fq2& B, const fq2& C) { return fq12{fq6{A, fq2_zero(), fq2_zero()}, fq6{B, C, fq2_zero()}}; }

// tangent at T (Jacobian over Fq2, twisted curve) evaluated at P = (xp, yp) in G1, scaled by subfield factors:
//   2 Y Z^3 yp  -  3 X^2 Z^2 xp w  +  (3 X^3 - 2 Y^2) w^3        (w^3 = v w)
ZKP_HD_NOINLINE inline fq12 line_double(const g2_jac& T, const fq& xp, const fq& yp) {
    const fq2 XX = f_sq(T.X), YY = f_sq(T.Y), ZZ = f_sq(T.Z);
    const fq2 A = fq2_mul_fq(f_dbl(f_mul(f_mul(T.Y, T.Z), ZZ)), yp);
    const fq2 x3 = f_add(f_dbl(XX), XX);                                  // 3 X^2
    const fq2 B = f_neg(fq2_mul_fq(f_mul(x3, ZZ), xp));
    const fq2 C = f_sub(f_mul(x3, T.X), f_dbl(YY));
    return fq12_from_line(A, B, C);
}
// chord through T (Jacobian) and Q (affine):  D yp - N xp w + (N x2 - D y2) w^3,  N = y2 Z^3 - Y,  D = (x2 Z^2 - X) Z
ZKP_HD_NOINLINE inline fq12 line_add(const g2_jac& T, const g2_aff& Q, const fq& xp, const fq& yp) {
    const fq2 ZZ = f_sq(T.Z);
    const fq2 N = f_sub(f_mul(f_mul(Q.y, T.Z), ZZ), T.Y);
    const fq2 D = f_mul(f_sub(f_mul(Q.x, ZZ), T.X), T.Z);
    return fq12_from_line(fq2_mul_fq(D, yp), f_neg(fq2_mul_fq(N, xp)), f_sub(f_mul(N, Q.x), f_mul(D, Q.y)));
}
// f_{t-1,Q}(P), t - 1 = 6 x^2 = 0x6f4d8248eeb859fbf83e9682e87cfd46 (127 bits)
ZKP_HD_NOINLINE inline fq12 miller_loop(const g2_aff& Q, const g1_aff& P) {
    const uint32_t T1[4] = {0xe87cfd46u, 0xf83e9682u, 0xeeb859fbu, 0x6f4d8248u};
    fq12 f = fq12_one();
    g2_jac T = jac_from_aff(Q);
    for (int i = 125; i >= 0; i--) {
        f = fq12_mul(fq12_sq(f), line_double(T, P.x, P.y));
        T = jac_dbl(T);
        if ((T1[i >> 5] >> (i & 31)) & 1u) {
            f = fq12_mul(f, line_add(T, Q, P.x, P.y));
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
ZKP_HD inline fq6 fq6_mul_fq2(const fq6& a, const fq2& k) { return fq6{f_mul(a.a0, k), f_mul(a.a1, k), f_mul(a.a2, k)}; }
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
