// BN254 G1 (over Fq) and G2 (over Fq2 = Fq[u]/(u^2+1)) in Jacobian coordinates, a = 0 short Weierstrass curves
// y^2 = x^3 + 3 and y^2 = x^3 + 3/(9+u).  Mixed additions against affine table entries are the inner loop of the
// Groth16 fixed-base MSMs (a_query, b_g1_query, h_query, l_query on G1; b_g2_query on G2).
// Replaces ark-ec's short_weierstrass Projective/Affine for ark-bn254 (used via ark-groth16 at
// /root/reference/src/backend/snark.rs:364,442); serialisation follows ark-serialize's uncompressed form.
#pragma once
#include "bn254_fq.h"
#include "bn254_fq9.h"

namespace zkp {

struct fq2 { fq c0, c1; };

// ---- uniform field interface so the point formulas are written once
// Fq through its "safe" operations (every result < 3p, carried); the MSM inner loop uses the lazy forms directly (g1_madd_lazy)
ZKP_HD inline fq f_add(const fq& a, const fq& b) { return fq_add(a, b); }
ZKP_HD inline fq f_sub(const fq& a, const fq& b) { return fq_sub(a, b); }
ZKP_HD inline fq f_mul(const fq& a, const fq& b) { return fq_mul(a, b); }
ZKP_HD inline fq f_sq(const fq& a) { return fq_sq(a); }
ZKP_HD inline fq f_neg(const fq& a) { return fq_neg(a); }
ZKP_HD inline fq f_dbl(const fq& a) { return fq_dbl(a); }
ZKP_HD inline bool f_is_zero(const fq& a) { return fq_is_zero(a); }
ZKP_HD inline fq f_select(bool c, const fq& a, const fq& b) { return fq_select(c, a, b); }
ZKP_HD inline void f_set_zero(fq& a) { a = fq_zero(); }
ZKP_HD inline void f_set_one(fq& a) { a = fq_one(); }
ZKP_HD inline fq f_inv(const fq& a) { return fq_inv(a); }

ZKP_HD inline fq2 f_add(const fq2& a, const fq2& b) { return fq2{fq_add(a.c0, b.c0), fq_add(a.c1, b.c1)}; }
ZKP_HD inline fq2 f_sub(const fq2& a, const fq2& b) { return fq2{fq_sub(a.c0, b.c0), fq_sub(a.c1, b.c1)}; }
ZKP_HD inline fq2 f_neg(const fq2& a) { return fq2{fq_neg(a.c0), fq_neg(a.c1)}; }
ZKP_HD inline fq2 f_dbl(const fq2& a) { return fq2{fq_dbl(a.c0), fq_dbl(a.c1)}; }
// Karatsuba, u^2 = -1.  Inputs safe (< 4p, carried); the three products take lazily added operands (< 8p, limbs < 2^27);
// c0 = t0 - t1 + 4p, c1 = t2 - t0 - t1 + 8p are reduced once each.
ZKP_HD inline fq2 f_mul(const fq2& a, const fq2& b) {
    const fq t0 = fq_mul(a.c0, b.c0), t1 = fq_mul(a.c1, b.c1);
    const fq t2 = fq_mul(fq_add_l(a.c0, a.c1), fq_add_l(b.c0, b.c1));
    return fq2{fq_reduce_weak(fq_sub_k4(t0, t1)), fq_reduce_weak(fq_sub_k8(t2, fq_add_l(t0, t1)))};
}
ZKP_HD inline fq2 f_sq(const fq2& a) {                           // (a0+a1)(a0-a1), 2 a0 a1
    const fq t = fq_mul(a.c0, a.c1);
    return fq2{fq_reduce_weak(fq_mul(fq_add_l(a.c0, a.c1), fq_sub_k4(a.c0, a.c1))), fq_reduce_weak(fq_dbl_l(t))};
}
ZKP_HD inline bool f_is_zero(const fq2& a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }
ZKP_HD inline fq2 f_select(bool c, const fq2& a, const fq2& b) { return fq2{fq_select(c, a.c0, b.c0), fq_select(c, a.c1, b.c1)}; }
ZKP_HD inline void f_set_zero(fq2& a) { a.c0 = fq_zero(); a.c1 = fq_zero(); }
ZKP_HD inline void f_set_one(fq2& a) { a.c0 = fq_one(); a.c1 = fq_zero(); }
ZKP_HD inline fq2 f_inv(const fq2& a) {
    const fq d = fq_inv(fq_add(fq_sq(a.c0), fq_sq(a.c1)));
    return fq2{fq_reduce_weak(fq_mul(a.c0, d)), fq_neg(fq_reduce_weak(fq_mul(a.c1, d)))};
}

template <class F> struct Aff { F x, y; };            // never the point at infinity (table entries, generators)
template <class F> struct Jac { F X, Y, Z; };         // Z == 0 <=> infinity
using g1_aff = Aff<fq>; using g1_jac = Jac<fq>;
using g2_aff = Aff<fq2>; using g2_jac = Jac<fq2>;

template <class F> ZKP_HD inline Jac<F> jac_infinity() { Jac<F> r; f_set_one(r.X); f_set_one(r.Y); f_set_zero(r.Z); return r; }
template <class F> ZKP_HD inline Jac<F> jac_from_aff(const Aff<F>& p) { Jac<F> r; r.X = p.x; r.Y = p.y; f_set_one(r.Z); return r; }
template <class F> ZKP_HD inline bool jac_is_inf(const Jac<F>& p) { return f_is_zero(p.Z); }

// dbl-2009-l (a = 0): 2M + 5S
template <class F> ZKP_HD inline Jac<F> jac_dbl(const Jac<F>& p) {
    const F A = f_sq(p.X), B = f_sq(p.Y), C = f_sq(B);
    const F D = f_dbl(f_sub(f_sub(f_sq(f_add(p.X, B)), A), C));
    const F E = f_add(f_dbl(A), A), Fq_ = f_sq(E);
    Jac<F> r;
    r.X = f_sub(Fq_, f_dbl(D));
    r.Y = f_sub(f_mul(E, f_sub(D, r.X)), f_dbl(f_dbl(f_dbl(C))));
    r.Z = f_dbl(f_mul(p.Y, p.Z));
    return r;                                           // Y == 0 never happens on these prime-order(-subgroup) curves
}

// madd-2007-bl (Jacobian + affine): 7M + 4S, with the exceptional cases handled (P = inf, P = +-Q)
template <class F> ZKP_HD inline Jac<F> jac_madd(const Jac<F>& p, const Aff<F>& q) {
    if (jac_is_inf(p)) return jac_from_aff(q);
    const F Z1Z1 = f_sq(p.Z);
    const F U2 = f_mul(q.x, Z1Z1), S2 = f_mul(f_mul(q.y, p.Z), Z1Z1);
    const F H = f_sub(U2, p.X), rr = f_dbl(f_sub(S2, p.Y));
    if (f_is_zero(H)) return f_is_zero(rr) ? jac_dbl(p) : jac_infinity<F>();
    const F HH = f_sq(H), I = f_dbl(f_dbl(HH)), J = f_mul(H, I), V = f_mul(p.X, I);
    Jac<F> r;
    r.X = f_sub(f_sub(f_sq(rr), J), f_dbl(V));
    r.Y = f_sub(f_mul(rr, f_sub(V, r.X)), f_dbl(f_mul(p.Y, J)));
    r.Z = f_sub(f_sub(f_sq(f_add(p.Z, H)), Z1Z1), HH);
    return r;
}

// The MSM inner loops use madd-2007-bl WITHOUT the exceptional-case branches (g1_madd_lazy, g2_madd_lazy): the
// accumulator starts from a fixed offset point (never infinity) and meeting +-(table entry) would need a discrete-log
// relation between key points and the offset (probability ~2^-250); H = 0 then collapses to Z3 = 0 rather than to
// wrong finite coordinates.  jac_madd above (with the branches) is the reference form the tests compare against.

// The G1 MSM inner loop: madd-2007-bl on lazily reduced limbs (bounds proved in tests/test_fq_bounds.py).
// p = (X1, Y1, Z1) safe (< 4p... in fact < 3p, carried); q affine with safe coordinates (table entries are canonical).
// With s = S2 - Y1 (so r = 2s):  X3 = 4 s^2 - J - 2V ;  Y3 = 2 (s (V - X3) - Y1 J) ;  Z3 = (Z1 + H)^2 - Z1Z1 - HH.
ZKP_HD inline Jac<fq> g1_madd_lazy(const Jac<fq>& p, const Aff<fq>& q) {
    const fq Z1Z1 = fq_sq(p.Z);
    const fq U2 = fq_mul(q.x, Z1Z1), S2 = fq_mul(fq_mul(q.y, p.Z), Z1Z1);
    const fq H = fq_sub_k4(U2, p.X), sv = fq_sub_k4(S2, p.Y);           // < 7p, limbs < 2^27.6
    const fq HH = fq_sq(H);
    const fq I = fq_dbl_l(fq_dbl_l(HH));                                 // 4 HH, limbs < 2^28
    const fq J = fq_mul(H, I), V = fq_mul(p.X, I);
    const fq ss = fq_sq(sv);
    Jac<fq> r;
    r.X = fq_reduce_weak(fq_sub_k8(fq_sub_k4(fq_dbl_l(fq_dbl_l(ss)), J), fq_dbl_l(V)));
    const fq t = fq_sub_k4(fq_mul(sv, fq_sub_k4(V, r.X)), fq_mul(p.Y, J));
    r.Y = fq_reduce_weak(fq_dbl_l(t));
    r.Z = fq_reduce_weak(fq_sub_k4(fq_sub_k4(fq_sq(fq_add_l(p.Z, H)), Z1Z1), HH));
    return r;
}

// The G1 MSM accumulator of round 2: extended Jacobian ("XYZZ") coordinates x = X / ZZ, y = Y / ZZZ with ZZ^3 = ZZZ^2.  A mixed
// addition (mmadd-2008-s) is 8 products + 2 squarings where madd-2007-bl needs 7 + 4, and its Y3 = R (Q - X3) - Y1 PPP is formed
// with ONE reduction (fq_mul_add2 on the lazily negated Y1): 1 810 limb multiply-adds per addition instead of 2 020.  Same
// no-exceptional-case convention as g1_madd_lazy (offset-point accumulators): P = 0 collapses to ZZ3 = ZZZ3 = 0, i.e. infinity.
struct g1_xyzz { fq X, Y, ZZ, ZZZ; };
ZKP_HD inline g1_xyzz g1_mmadd_lazy(const g1_xyzz& p, const Aff<fq>& q) {
    const fq U2 = fq_mul(q.x, p.ZZ), S2 = fq_mul(q.y, p.ZZZ);
    const fq P = fq_sub_k4(U2, p.X), Rv = fq_sub_k4(S2, p.Y);             // < 7p, limbs < 2^27.6
    const fq PP = fq_sq(P);
    const fq PPP = fq_mul(P, PP), Q = fq_mul(p.X, PP);
    const fq RR = fq_sq(Rv);
    g1_xyzz r;
    r.X = fq_reduce_weak(fq_sub_k8(fq_sub_k4(RR, PPP), fq_dbl_l(Q)));
    r.Y = fq_mul_add2(Rv, fq_sub_k4(Q, r.X), fq_sub_k4(fq_zero(), p.Y), PPP);   // R (Q - X3) + (4p - Y1) PPP, one reduction
    r.ZZ = fq_mul(p.ZZ, PP);
    r.ZZZ = fq_mul(p.ZZZ, PPP);
    return r;
}
// The same addition on nine 29-bit limbs (bn254_fq9.h), the form k_msm_gather<G1Msm> runs in.  Value bounds in units of p, every
// operand carried9: accumulator X < 8, Y < 4, ZZ, ZZZ < 2; table entry x, y < 4.  U2, S2 < 4*2/169+1 = 1.05; P < 9.05, R < 5.05;
// PP < 1.49, PPP < 1.08, Q < 1.08, RR < 1.16; X3 < 5.16; Y3 < (5.05*9.08 + 4*1.08)/169 + 1 = 1.30; ZZ3, ZZZ3 < 1.02.
struct g1_xyzz9 { fq9 X, Y, ZZ, ZZZ; };
struct g1_aff9 { fq9 x, y; };
ZKP_HD inline g1_xyzz9 g1_mmadd9(const g1_xyzz9& p, const g1_aff9& q) {
    const fq9 U2 = fq9_mul(q.x, p.ZZ), S2 = fq9_mul(q.y, p.ZZZ);
    const fq9 P = fq9_sub_k<8>(U2, p.X), Rv = fq9_sub_k<4>(S2, p.Y);
    const fq9 PP = fq9_sq(P);
    const fq9 PPP = fq9_mul(P, PP), Q = fq9_mul(p.X, PP);
    const fq9 RR = fq9_sq(Rv);
    g1_xyzz9 r;
    r.X = fq9_sub2_k4(RR, PPP, Q);                                       // R^2 - PPP - 2 Q + 4p
    r.Y = fq9_mul_add2(Rv, fq9_sub_loose<8>(Q, r.X), fq9_neg_loose<4>(p.Y), PPP);      // R (Q - X3 + 8p) + (4p - Y1) PPP, one reduction; the two differences stay loose (bn254_fq9.h)
    r.ZZ = fq9_mul(p.ZZ, PP);
    r.ZZZ = fq9_mul(p.ZZZ, PPP);
    return r;
}
ZKP_HD inline g1_xyzz9 xyzz9_from_jac(const Jac<fq>& p) {
    const fq zz = fq_sq(p.Z);
    return g1_xyzz9{fq9_from_fq(p.X), fq9_from_fq(p.Y), fq9_from_fq(zz), fq9_from_fq(fq_mul(zz, p.Z))};
}
ZKP_HD inline Jac<fq> jac_from_xyzz9(const g1_xyzz9& p) { return Jac<fq>{fq9_to_fq(fq9_mul(p.X, p.ZZ)), fq9_to_fq(fq9_mul(p.Y, p.ZZZ)), fq9_to_fq(p.ZZ)}; }
// (X, Y, Z) Jacobian <-> XYZZ: ZZ = Z^2, ZZZ = Z^3 one way; the other way Z' = ZZ gives X' = X ZZ, Y' = Y ZZZ
ZKP_HD inline g1_xyzz xyzz_from_jac(const Jac<fq>& p) { const fq zz = fq_sq(p.Z); return g1_xyzz{p.X, p.Y, zz, fq_mul(zz, p.Z)}; }
ZKP_HD inline Jac<fq> jac_from_xyzz(const g1_xyzz& p) { return Jac<fq>{fq_reduce_weak(fq_mul(p.X, p.ZZ)), fq_reduce_weak(fq_mul(p.Y, p.ZZZ)), fq_reduce_weak(p.ZZ)}; }

// The G2 MSM inner loop: the same formulas over Fq2 with the Karatsuba products kept apart (t0 = a0 b0, t1 = a1 b1,
// t2 = (a0+a1)(b0+b1); c0 = t0 - t1, c1 = t2 - t0 - t1), so that every sum of products is formed limb-wise from carried
// product outputs and reduced once: 10 weak reductions per mixed addition instead of ~50 with the always-reduced forms.
struct fq2k { fq t0, t1, t2; };
struct fq2s { fq c0, m; };                             // a^2 = (c0, 2 m)
ZKP_HD inline fq2k fq2_kara(const fq2& a, const fq2& b) { return fq2k{fq_mul(a.c0, b.c0), fq_mul(a.c1, b.c1), fq_mul(fq_add_l(a.c0, a.c1), fq_add_l(b.c0, b.c1))}; }
ZKP_HD inline fq2 fq2_join(const fq2k& k) { return fq2{fq_sub_k4(k.t0, k.t1), fq_sub_k8(k.t2, fq_add_l(k.t0, k.t1))}; }   // unreduced, uncarried
ZKP_HD inline fq2s fq2_sq_l(const fq2& a) { return fq2s{fq_mul(fq_add_l(a.c0, a.c1), fq_sub_k4(a.c0, a.c1)), fq_mul(a.c0, a.c1)}; }   // a safe
ZKP_HD inline Jac<fq2> g2_madd_lazy(const Jac<fq2>& p, const Aff<fq2>& q, bool negate) {
    const fq zero = fq_zero();
    const fq2 y2{fq_select(negate, fq_sub_k4(zero, q.y.c0), q.y.c0), fq_select(negate, fq_sub_k4(zero, q.y.c1), q.y.c1)};
    const fq2s zz = fq2_sq_l(p.Z);
    const fq2 Z1Z1{zz.c0, fq_dbl_l(zz.m)};
    const fq2 U2 = fq2_join(fq2_kara(q.x, Z1Z1));
    const fq2 S2 = fq2_join(fq2_kara(fq2_join(fq2_kara(y2, p.Z)), Z1Z1));
    const fq2 H{fq_reduce_weak(fq_sub_k4(U2.c0, p.X.c0)), fq_reduce_weak(fq_sub_k4(U2.c1, p.X.c1))};
    const fq2 sv{fq_reduce_weak(fq_sub_k4(S2.c0, p.Y.c0)), fq_reduce_weak(fq_sub_k4(S2.c1, p.Y.c1))};
    const fq2s hh = fq2_sq_l(H);
    const fq2 I{fq_dbl_l(fq_dbl_l(hh.c0)), fq_dbl_l(fq_dbl_l(fq_dbl_l(hh.m)))};                  // 4 HH
    const fq2k J = fq2_kara(H, I), V = fq2_kara(p.X, I);
    const fq2s ss = fq2_sq_l(sv);
    Jac<fq2> r;
    // X3 = 4 s^2 - J - 2V
    r.X.c0 = fq_reduce_weak(fq_sub_k16(fq_add_l(fq_dbl_l(fq_dbl_l(ss.c0)), fq_add_l(J.t1, fq_dbl_l(V.t1))), fq_add_l(J.t0, fq_dbl_l(V.t0))));
    r.X.c1 = fq_reduce_weak(fq_sub_k16(fq_add_l(fq_dbl_l(fq_dbl_l(fq_dbl_l(ss.m))), fq_add_l(fq_add_l(J.t0, J.t1), fq_dbl_l(fq_add_l(V.t0, V.t1)))),
                                       fq_add_l(J.t2, fq_dbl_l(V.t2))));
    // Y3 = 2 (s (V - X3) - Y1 J)
    const fq2 W{fq_sub_k8(V.t0, fq_add_l(V.t1, r.X.c0)), fq_sub_k16(V.t2, fq_add_l(fq_add_l(V.t0, V.t1), r.X.c1))};
    const fq2k P1 = fq2_kara(sv, W), P2 = fq2_kara(p.Y, fq2_join(J));
    r.Y.c0 = fq_reduce_weak(fq_dbl_l(fq_sub_k8(fq_add_l(P1.t0, P2.t1), fq_add_l(P1.t1, P2.t0))));
    r.Y.c1 = fq_reduce_weak(fq_dbl_l(fq_sub_k16(fq_add_l(P1.t2, fq_add_l(P2.t0, P2.t1)), fq_add_l(fq_add_l(P1.t0, P1.t1), P2.t2))));
    // Z3 = (Z1 + H)^2 - Z1Z1 - HH
    const fq2 zh{fq_add_l(p.Z.c0, H.c0), fq_add_l(p.Z.c1, H.c1)};
    const fq zq0 = fq_mul(fq_add_l(zh.c0, zh.c1), fq_sub_k8(zh.c0, zh.c1)), zqm = fq_mul(zh.c0, zh.c1);
    r.Z.c0 = fq_reduce_weak(fq_sub_k8(zq0, fq_add_l(zz.c0, hh.c0)));
    r.Z.c1 = fq_reduce_weak(fq_dbl_l(fq_sub_k8(zqm, fq_add_l(zz.m, hh.m))));
    return r;
}

// The G2 MSM loop on nine 29-bit limbs: XYZZ coordinates over Fq2 = Fq[u]/(u^2 + 1) (mmadd-2008-s).  An Fq2 product is two
// fused double products (a0 b0 + (Kp - a1) b1, a0 b1 + a1 b0: 486 multiply-adds, two reductions, one negation pass that several
// products share); Y3 = R (Q - X3) - Y1 PPP is two fused quadruple products.  4 536 multiply-adds per addition where the
// Karatsuba form on ten limbs (g2_madd_lazy, the host reference) takes 5 800.  Value bounds (units of p, every operand carried9;
// tests/test_fq_bounds.py): X < 10.4, Y < 4, ZZ, ZZZ < 3 (< 2 after the first addition), entries < 3.
struct fq2_9 { fq9 c0, c1; };
struct g2_xyzz9 { fq2_9 X, Y, ZZ, ZZZ; };
struct g2_aff9 { fq2_9 x, y; };
// a * b with nb1 = K p - b.c1 supplied by the caller
ZKP_HD inline fq2_9 fq2_9_mul(const fq2_9& a, const fq2_9& b, const fq9& nb1) {
    return fq2_9{fq9_mul_add2(a.c0, b.c0, a.c1, nb1), fq9_mul_add2(a.c0, b.c1, a.c1, b.c0)};
}
// a^2 with na1 = K p - a.c1
ZKP_HD inline fq2_9 fq2_9_sq(const fq2_9& a, const fq9& na1) { return fq2_9{fq9_mul_add2(a.c0, a.c0, a.c1, na1), fq9_mul(fq9_dbl_l(a.c0), a.c1)}; }
ZKP_HD inline g2_xyzz9 g2_mmadd9(const g2_xyzz9& p, const g2_aff9& q, bool negate) {
    const fq9 nZZ1 = fq9_neg_k<4>(p.ZZ.c1), nZZZ1 = fq9_neg_k<4>(p.ZZZ.c1);
    const fq2_9 U2 = fq2_9_mul(q.x, p.ZZ, nZZ1), S2 = fq2_9_mul(q.y, p.ZZZ, nZZZ1);                   // < 1.11
    const fq2_9 P{fq9_sub_k<16>(U2.c0, p.X.c0), fq9_sub_k<16>(U2.c1, p.X.c1)};                        // < 17.2
    const fq2_9 Rv{fq9_sgn_sub_k<8>(negate, S2.c0, p.Y.c0), fq9_sgn_sub_k<8>(negate, S2.c1, p.Y.c1)}; // +-S2 - Y1 + 8p < 9.2
    const fq2_9 PP = fq2_9_sq(P, fq9_neg_k<32>(P.c1));                                                // < 6
    const fq9 nPP1 = fq9_neg_k<8>(PP.c1);
    const fq2_9 PPP = fq2_9_mul(P, PP, nPP1), Q = fq2_9_mul(p.X, PP, nPP1);                           // < 2.8, < 2.0
    const fq9 nR1 = fq9_neg_k<16>(Rv.c1);
    const fq2_9 RR = fq2_9_sq(Rv, nR1);                                                               // < 2.4
    g2_xyzz9 r;
    r.X = fq2_9{fq9_sub2_k<8>(RR.c0, PPP.c0, Q.c0), fq9_sub2_k<8>(RR.c1, PPP.c1, Q.c1)};              // R^2 - PPP - 2Q + 8p < 10.4
    const fq2_9 W{fq9_sub_k<16>(Q.c0, r.X.c0), fq9_sub_k<16>(Q.c1, r.X.c1)};                          // < 18
    const fq9 nY0 = fq9_neg_k<4>(p.Y.c0), nY1 = fq9_neg_k<4>(p.Y.c1);
    r.Y = fq2_9{fq9_mul_add4(Rv.c0, W.c0, nR1, W.c1, nY0, PPP.c0, p.Y.c1, PPP.c1),                    // Re(R W) - Re(Y1 PPP)
                fq9_mul_add4(Rv.c0, W.c1, Rv.c1, W.c0, nY0, PPP.c1, nY1, PPP.c0)};                    // Im(R W) - Im(Y1 PPP)
    r.ZZ = fq2_9_mul(PP, p.ZZ, nZZ1);
    r.ZZZ = fq2_9_mul(PPP, p.ZZZ, nZZZ1);
    return r;
}
ZKP_HD inline fq2_9 fq2_9_from_fq2(const fq2& a) { return fq2_9{fq9_from_fq(a.c0), fq9_from_fq(a.c1)}; }
ZKP_HD inline fq2 fq2_9_to_fq2(const fq2_9& a) { return fq2{fq9_to_fq(a.c0), fq9_to_fq(a.c1)}; }
ZKP_HD inline g2_xyzz9 g2_xyzz9_from_jac(const Jac<fq2>& p) {
    const fq2 zz = f_sq(p.Z);
    return g2_xyzz9{fq2_9_from_fq2(p.X), fq2_9_from_fq2(p.Y), fq2_9_from_fq2(zz), fq2_9_from_fq2(f_mul(zz, p.Z))};
}
ZKP_HD inline Jac<fq2> jac_from_g2_xyzz9(const g2_xyzz9& p) {      // Z' = ZZ: X' = X ZZ, Y' = Y ZZZ
    return Jac<fq2>{fq2_9_to_fq2(fq2_9_mul(p.X, p.ZZ, fq9_neg_k<4>(p.ZZ.c1))), fq2_9_to_fq2(fq2_9_mul(p.Y, p.ZZZ, fq9_neg_k<4>(p.ZZZ.c1))), fq2_9_to_fq2(p.ZZ)};
}

// add-2007-bl (Jacobian + Jacobian): 11M + 5S
template <class F> ZKP_HD inline Jac<F> jac_add(const Jac<F>& p, const Jac<F>& q) {
    if (jac_is_inf(p)) return q;
    if (jac_is_inf(q)) return p;
    const F Z1Z1 = f_sq(p.Z), Z2Z2 = f_sq(q.Z);
    const F U1 = f_mul(p.X, Z2Z2), U2 = f_mul(q.X, Z1Z1);
    const F S1 = f_mul(f_mul(p.Y, q.Z), Z2Z2), S2 = f_mul(f_mul(q.Y, p.Z), Z1Z1);
    const F H = f_sub(U2, U1), rr = f_dbl(f_sub(S2, S1));
    if (f_is_zero(H)) return f_is_zero(rr) ? jac_dbl(p) : jac_infinity<F>();
    const F I = f_sq(f_dbl(H)), J = f_mul(H, I), V = f_mul(U1, I);
    Jac<F> r;
    r.X = f_sub(f_sub(f_sq(rr), J), f_dbl(V));
    r.Y = f_sub(f_mul(rr, f_sub(V, r.X)), f_dbl(f_mul(S1, J)));
    r.Z = f_mul(f_sub(f_sub(f_sq(f_add(p.Z, q.Z)), Z1Z1), Z2Z2), H);
    return r;
}
template <class F> ZKP_HD inline Aff<F> aff_neg(const Aff<F>& p) { return Aff<F>{p.x, f_neg(p.y)}; }
template <class F> ZKP_HD inline Jac<F> jac_neg(const Jac<F>& p) { return Jac<F>{p.X, f_neg(p.Y), p.Z}; }

// variable-base scalar multiplication, 4-bit fixed window over a raw 256-bit scalar (s*A and r*B1 of the Groth16 C element)
template <class F> ZKP_HD inline Jac<F> jac_mul_raw(const Jac<F>& p, const uint32_t k[8]) {
    Jac<F> tbl[16];
    tbl[0] = jac_infinity<F>(); tbl[1] = p;
    for (int i = 2; i < 16; i++) tbl[i] = jac_add(tbl[i - 1], p);
    Jac<F> acc = jac_infinity<F>();
    for (int nib = 63; nib >= 0; nib--) {
        acc = jac_dbl(jac_dbl(jac_dbl(jac_dbl(acc))));
        const uint32_t d = (k[nib >> 3] >> ((nib & 7) * 4)) & 15u;
        if (d) acc = jac_add(acc, tbl[d]);
    }
    return acc;
}

// ---- GLV split for G1 (s*A and r*B1 of the Groth16 C element).  BN254 has the endomorphism phi(x, y) = (beta x, y) = lambda (x, y)
// with beta^2 + beta + 1 = 0 in Fq and lambda^2 + lambda + 1 = 0 in Fr, so k P = k1 P + k2 phi(P) with |k1|, |k2| < 2^128
// (k = k1 + k2 lambda mod r): two half-length multiplications on two lanes instead of one 254-bit ladder on one -- a lone wave gets
// half of a SIMD's multiply-add rate however its instructions are arranged (profiles/r02_mad_ilp.txt), so latency only falls by
// spreading a proof over more lanes.  Lattice basis of {(a, b): a + b lambda = 0 mod r} from the extended Euclidean algorithm on
// (r, lambda): v1 = (A1, -NB1), v2 = (A2, B2), det = r; c1 = floor(k G1 / 2^256) ~ B2 k / r, c2 = floor(k G2 / 2^256) ~ NB1 k / r,
// k1 = k - c1 A1 - c2 A2, k2 = c1 NB1 - c2 B2, both taken mod 2^160 as two's-complement numbers (exhaustively sampled in
// tests/test_emul_groth16.py: magnitudes stay below 2^128 with the floor quotients).
template <int NA, int NB, int NO> ZKP_HD inline void mp_mul_lo(uint32_t* out, const uint32_t* a, const uint32_t* b) {      // low NO words of a * b
    uint64_t acc = 0, hi = 0;
    ZKP_UNROLL for (int k = 0; k < NO; k++) {
        ZKP_UNROLL for (int i = 0; i < NA; i++) {
            const int j = k - i;
            if (j < 0 || j >= NB) continue;
            const uint64_t t = (uint64_t)a[i] * b[j];
            acc += (uint32_t)t; hi += t >> 32;
        }
        out[k] = (uint32_t)acc;
        acc = (acc >> 32) + (uint32_t)hi; hi >>= 32;
    }
}
struct glv_half { uint32_t mag[4]; bool neg; };
ZKP_HD inline glv_half glv_from_160(const uint32_t w[5]) {       // two's complement mod 2^160, |value| < 2^128
    glv_half h; h.neg = (w[4] >> 31) != 0;
    uint32_t c = h.neg ? 1u : 0u;
    ZKP_UNROLL for (int i = 0; i < 4; i++) { const uint32_t x = h.neg ? ~w[i] : w[i]; const uint32_t t = x + c; c = (c && t == 0) ? 1u : 0u; h.mag[i] = t; }
    return h;
}
ZKP_HD inline void fr_glv_split(const uint32_t k[8], glv_half& h1, glv_half& h2) {
    const uint32_t G1[3] = {0xc7e0b3d7u, 0xd91d232eu, 0x00000002u}, G2[5] = {0x391eb18du, 0x7a7bd9d4u, 0xa773d2cfu, 0x4ccef014u, 0x00000002u};
    const uint32_t A1[2] = {0x94d213e3u, 0x89d32568u}, A2[4] = {0x1221250bu, 0x0be4e154u, 0xeeb859fdu, 0x6f4d8248u};
    const uint32_t NB1[4] = {0x7d4f1128u, 0x8211bbebu, 0xeeb859fcu, 0x6f4d8248u}, B2[2] = {0x94d213e3u, 0x89d32568u};
    uint32_t t1[11], t2[13];
    mp_mul_lo<8, 3, 11>(t1, k, G1); mp_mul_lo<8, 5, 13>(t2, k, G2);
    const uint32_t* c1 = t1 + 8;      // 3 words
    const uint32_t* c2 = t2 + 8;      // 5 words
    uint32_t p11[5], p22[5], p12[5], p21[5];
    mp_mul_lo<3, 2, 5>(p11, c1, A1); mp_mul_lo<5, 4, 5>(p22, c2, A2); mp_mul_lo<3, 4, 5>(p12, c1, NB1); mp_mul_lo<5, 2, 5>(p21, c2, B2);
    uint32_t k1[5], k2[5]; int64_t b1 = 0, b2 = 0;
    ZKP_UNROLL for (int i = 0; i < 5; i++) {
        b1 += (int64_t)k[i] - (int64_t)p11[i] - (int64_t)p22[i]; k1[i] = (uint32_t)b1; b1 >>= 32;
        b2 += (int64_t)p12[i] - (int64_t)p21[i]; k2[i] = (uint32_t)b2; b2 >>= 32;
    }
    h1 = glv_from_160(k1); h2 = glv_from_160(k2);
}
ZKP_HD inline fq fq_glv_beta() {       // beta with phi(P) = lambda P for the lambda of fr_glv_split
    const uint32_t w[8] = {0x77fffffeu, 0x57634731u, 0xacdb5c4fu, 0xd4f263f1u, 0xa0d48bacu, 0x59e26bceu, 0x00000000u, 0x00000000u};
    return fq_from_raw(w);
}
// (+-mag) * p for a magnitude below 2^128: signed 4-bit windows (33 digits in [-8, 8]) over the table p, 2p, ..., 8p
template <class F> ZKP_HD inline Jac<F> jac_mul_u128_signed(const Jac<F>& p, const uint32_t mag[4], bool neg) {
    Jac<F> tbl[8];
    tbl[0] = p; tbl[1] = jac_dbl(p);
    for (int i = 2; i < 8; i++) tbl[i] = jac_add(tbl[i - 1], p);
    int dig[33]; int carry = 0;
    for (int i = 0; i < 32; i++) {
        int v = (int)((mag[i >> 3] >> ((i & 7) * 4)) & 15u) + carry;
        carry = v > 8 ? 1 : 0; if (carry) v -= 16;
        dig[i] = v;
    }
    dig[32] = carry;
    Jac<F> acc = jac_infinity<F>();
    for (int i = 32; i >= 0; i--) {
        if (i != 32) acc = jac_dbl(jac_dbl(jac_dbl(jac_dbl(acc))));
        const int d = dig[i];
        if (d > 0) acc = jac_add(acc, tbl[d - 1]);
        else if (d < 0) acc = jac_add(acc, jac_neg(tbl[-d - 1]));
    }
    return neg ? jac_neg(acc) : acc;
}

// (x, y) affine; returns false for the point at infinity
template <class F> ZKP_HD inline bool jac_to_aff(Aff<F>& out, const Jac<F>& p) {
    if (jac_is_inf(p)) return false;
    const F zi = f_inv(p.Z), zi2 = f_sq(zi);
    out.x = f_mul(p.X, zi2); out.y = f_mul(f_mul(p.Y, zi2), zi);
    return true;
}

// ---- ark-serialize uncompressed: LE canonical coordinates, flags in the top two bits of the last byte
ZKP_HD inline bool fq_raw_gt_half(const uint32_t w[8]) {   // w > (p-1)/2  <=>  y is the lexicographically larger of {y, -y}
    const uint32_t half[8] = {0x6c3e7ea3u, 0x9e10460bu, 0xb438e546u, 0xcbc0b548u, 0x40c0ac2eu, 0xdc2822dbu, 0x7098d014u, 0x18322739u};
    for (int i = 7; i >= 0; i--) { if (w[i] > half[i]) return true; if (w[i] < half[i]) return false; }
    return false;
}
ZKP_HD inline void g1_serialize(uint32_t out[16], const g1_jac& p) {
    g1_aff a;
    if (!jac_to_aff(a, p)) { for (int i = 0; i < 16; i++) out[i] = 0; out[15] = 0x40000000u; return; }
    fq_to_raw(out, a.x); fq_to_raw(out + 8, a.y);
    if (fq_raw_gt_half(out + 8)) out[15] |= 0x80000000u;
}
ZKP_HD inline void g2_serialize(uint32_t out[32], const g2_jac& p) {
    g2_aff a;
    if (!jac_to_aff(a, p)) { for (int i = 0; i < 32; i++) out[i] = 0; out[31] = 0x40000000u; return; }
    fq_to_raw(out, a.x.c0); fq_to_raw(out + 8, a.x.c1); fq_to_raw(out + 16, a.y.c0); fq_to_raw(out + 24, a.y.c1);
    uint32_t nz = 0; for (int i = 0; i < 8; i++) nz |= out[24 + i];
    const bool larger = nz ? fq_raw_gt_half(out + 24) : fq_raw_gt_half(out + 16);   // order: c1 first, then c0
    if (larger) out[31] |= 0x80000000u;
}

}  // namespace zkp
