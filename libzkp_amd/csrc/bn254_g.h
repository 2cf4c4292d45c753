// BN254 G1 (over Fq) and G2 (over Fq2 = Fq[u]/(u^2+1)) in Jacobian coordinates, a = 0 short Weierstrass curves
// y^2 = x^3 + 3 and y^2 = x^3 + 3/(9+u).  Mixed additions against affine table entries are the inner loop of the
// Groth16 fixed-base MSMs (a_query, b_g1_query, h_query, l_query on G1; b_g2_query on G2).
// Replaces ark-ec's short_weierstrass Projective/Affine for ark-bn254 (used via ark-groth16 at
// /root/reference/src/backend/snark.rs:364,442); serialisation follows ark-serialize's uncompressed form.
#pragma once
#include "bn254_fp.h"

namespace zkp {

struct fq2 { fq c0, c1; };

// ---- uniform field interface so the point formulas are written once
ZKP_HD inline fq f_add(const fq& a, const fq& b) { return fp_add(a, b); }
ZKP_HD inline fq f_sub(const fq& a, const fq& b) { return fp_sub(a, b); }
ZKP_HD inline fq f_mul(const fq& a, const fq& b) { return fp_mul(a, b); }
ZKP_HD inline fq f_sq(const fq& a) { return fp_sq(a); }
ZKP_HD inline fq f_neg(const fq& a) { return fp_neg(a); }
ZKP_HD inline fq f_dbl(const fq& a) { return fp_dbl(a); }
ZKP_HD inline bool f_is_zero(const fq& a) { return fp_is_zero(a); }
ZKP_HD inline fq f_select(bool c, const fq& a, const fq& b) { return fp_select(c, a, b); }
ZKP_HD inline void f_set_zero(fq& a) { a = fp_zero<FqParams>(); }
ZKP_HD inline void f_set_one(fq& a) { a = fp_one<FqParams>(); }
ZKP_HD inline fq f_inv(const fq& a) { return fp_inv(a); }

ZKP_HD inline fq2 f_add(const fq2& a, const fq2& b) { return fq2{fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; }
ZKP_HD inline fq2 f_sub(const fq2& a, const fq2& b) { return fq2{fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; }
ZKP_HD inline fq2 f_neg(const fq2& a) { return fq2{fp_neg(a.c0), fp_neg(a.c1)}; }
ZKP_HD inline fq2 f_dbl(const fq2& a) { return fq2{fp_dbl(a.c0), fp_dbl(a.c1)}; }
ZKP_HD inline fq2 f_mul(const fq2& a, const fq2& b) {            // Karatsuba, u^2 = -1
    const fq t0 = fp_mul(a.c0, b.c0), t1 = fp_mul(a.c1, b.c1);
    const fq t2 = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
    return fq2{fp_sub(t0, t1), fp_sub(fp_sub(t2, t0), t1)};
}
ZKP_HD inline fq2 f_sq(const fq2& a) {                           // (a0+a1)(a0-a1), 2 a0 a1
    const fq t = fp_mul(a.c0, a.c1);
    return fq2{fp_mul(fp_add(a.c0, a.c1), fp_sub(a.c0, a.c1)), fp_dbl(t)};
}
ZKP_HD inline bool f_is_zero(const fq2& a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
ZKP_HD inline fq2 f_select(bool c, const fq2& a, const fq2& b) { return fq2{fp_select(c, a.c0, b.c0), fp_select(c, a.c1, b.c1)}; }
ZKP_HD inline void f_set_zero(fq2& a) { a.c0 = fp_zero<FqParams>(); a.c1 = fp_zero<FqParams>(); }
ZKP_HD inline void f_set_one(fq2& a) { a.c0 = fp_one<FqParams>(); a.c1 = fp_zero<FqParams>(); }
ZKP_HD inline fq2 f_inv(const fq2& a) {
    const fq d = fp_inv(fp_add(fp_sq(a.c0), fp_sq(a.c1)));
    return fq2{fp_mul(a.c0, d), fp_neg(fp_mul(a.c1, d))};
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
    fp_to_raw(out, a.x); fp_to_raw(out + 8, a.y);
    if (fq_raw_gt_half(out + 8)) out[15] |= 0x80000000u;
}
ZKP_HD inline void g2_serialize(uint32_t out[32], const g2_jac& p) {
    g2_aff a;
    if (!jac_to_aff(a, p)) { for (int i = 0; i < 32; i++) out[i] = 0; out[31] = 0x40000000u; return; }
    fp_to_raw(out, a.x.c0); fp_to_raw(out + 8, a.x.c1); fp_to_raw(out + 16, a.y.c0); fp_to_raw(out + 24, a.y.c1);
    uint32_t nz = 0; for (int i = 0; i < 8; i++) nz |= out[24 + i];
    const bool larger = nz ? fq_raw_gt_half(out + 24) : fq_raw_gt_half(out + 16);   // order: c1 first, then c0
    if (larger) out[31] |= 0x80000000u;
}

}  // namespace zkp
