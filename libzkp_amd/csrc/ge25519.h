// edwards25519 / ristretto255 group operations on the 10x25.5 field representation.
// Replaces curve25519-dalek's EdwardsPoint/RistrettoPoint (+ compress) as used at
// /root/reference/src/backend/bulletproofs.rs:4,134 and inside bulletproofs' prover.
// Encode follows RFC 9496 section 4.3.2; the one-way map (host side, generator derivation) section 4.3.4.
#pragma once
#include "fe25519.h"

namespace zkp {

struct ge { fe X, Y, Z, T; };             // extended coordinates, a = -1, all limbs carried
struct ge_niels { fe ypx, ymx, xy2d; };   // affine precomputed: y+x, y-x, 2dxy (limbs carried / canonical)

ZKP_HD inline fe fe_const_d2() { const uint32_t w[8] = {0x26b2f159u, 0xebd69b94u, 0x8283b156u, 0x00e0149au, 0xeef3d130u, 0x198e80f2u, 0x56dffce7u, 0x2406d9dcu}; return fe_fromwords(w); }
ZKP_HD inline fe fe_const_invsqrt_a_minus_d() { const uint32_t w[8] = {0x805d40eau, 0x99c8fdaau, 0x5a4172beu, 0x9d2f1617u, 0xfe01d840u, 0x16c27b91u, 0xcfaffca2u, 0x786c8905u}; return fe_fromwords(w); }
ZKP_HD inline fe fe_const_sqrt_ad_minus_one() { const uint32_t w[8] = {0x497b2e1bu, 0x7e97f6a0u, 0x1b7854bdu, 0xaf9d8e0cu, 0x31f5d1fdu, 0x0f3cfcc9u, 0x2b8348acu, 0x376931bfu}; return fe_fromwords(w); }
ZKP_HD inline fe fe_const_one_minus_d_sq() { const uint32_t w[8] = {0x945fc176u, 0xe27c09c1u, 0xcd5e350fu, 0x2c81a138u, 0xbe70dfe4u, 0x9994abddu, 0xb2b3e0d7u, 0x029072a8u}; return fe_fromwords(w); }
ZKP_HD inline fe fe_const_d_minus_one_sq() { const uint32_t w[8] = {0x44ed4d20u, 0x31ad5aaau, 0xb01e1999u, 0xd29e4a2cu, 0x529b4eebu, 0x4cdcd32fu, 0xf66c2241u, 0x5968b37au}; return fe_fromwords(w); }

ZKP_HD inline ge ge_identity() { ge r; r.X = fe_zero(); r.Y = fe_one(); r.Z = fe_one(); r.T = fe_zero(); return r; }

// shared tail of the unified addition (add-2008-hwcd-3): given A, B, C (carried) and D = 2*Z1*Z2 (< 2^27 even limbs)
ZKP_HD inline ge ge_finish_add(const fe& A, const fe& B, const fe& C, const fe& D) {
    const fe E = fe_sub(B, A);   // < 1.5*2^27
    const fe F = fe_sub(D, C);   // loose (< 2^28)
    const fe G = fe_add(D, C);   // < 1.5*2^27
    const fe H = fe_add(B, A);   // < 2^27
    ge r;
    r.X = fe_mul(F, E);
    r.Y = fe_mul(G, H);
    r.Z = fe_mul(F, G);
    r.T = fe_mul(E, H);
    return r;
}

// p + q, q affine niels (7 field multiplications)
ZKP_HD inline ge ge_madd(const ge& p, const ge_niels& q) {
    const fe A = fe_mul(fe_sub(p.Y, p.X), q.ymx);
    const fe B = fe_mul(fe_add(p.Y, p.X), q.ypx);
    const fe C = fe_mul(p.T, q.xy2d);
    const fe D = fe_add(p.Z, p.Z);
    return ge_finish_add(A, B, C, D);
}

ZKP_HD inline ge_niels ge_niels_neg(const ge_niels& q) { ge_niels r; r.ypx = q.ymx; r.ymx = q.ypx; r.xy2d = fe_neg(q.xy2d); return r; }
ZKP_HD inline ge_niels ge_niels_select(bool c, const ge_niels& a, const ge_niels& b) {
    ge_niels r; r.ypx = fe_select(c, a.ypx, b.ypx); r.ymx = fe_select(c, a.ymx, b.ymx); r.xy2d = fe_select(c, a.xy2d, b.xy2d); return r;
}

// p + q, both extended (9 field multiplications)
ZKP_HD inline ge ge_add(const ge& p, const ge& q) {
    const fe A = fe_mul(fe_sub(p.Y, p.X), fe_sub(q.Y, q.X));
    const fe B = fe_mul(fe_add(p.Y, p.X), fe_add(q.Y, q.X));
    const fe C = fe_mul(fe_mul(p.T, q.T), fe_const_d2());
    const fe zz = fe_mul(p.Z, q.Z);
    const fe D = fe_add(zz, zz);
    return ge_finish_add(A, B, C, D);
}

ZKP_HD inline ge ge_neg(const ge& p) { ge r; r.X = fe_carry(fe_neg(p.X)); r.Y = p.Y; r.Z = p.Z; r.T = fe_carry(fe_neg(p.T)); return r; }

// 2p (dbl-2008-hwcd); used for table construction only -- the prover itself needs no doublings
ZKP_HD inline ge ge_dbl(const ge& p) {
    const fe A = fe_sq(p.X), B = fe_sq(p.Y), zz = fe_sq(p.Z);
    const fe C = fe_add(zz, zz);
    const fe xy = fe_add(p.X, p.Y);
    const fe t = fe_sq(xy);
    const fe Hn = fe_add(A, B);                 // -H
    const fe En = fe_sub(Hn, t);                // -E
    const fe Gn = fe_sub(A, B);                 // -G
    const fe Fn = fe_carry(fe_add(C, Gn));      // -F
    ge r;
    r.X = fe_mul(En, Fn);
    r.Y = fe_mul(Gn, Hn);
    r.Z = fe_mul(Fn, Gn);
    r.T = fe_mul(En, Hn);
    return r;
}

// RFC 9496 4.3.2: canonical 32-byte ristretto255 encoding as eight little-endian words
ZKP_HD inline void ge_ristretto_encode(uint32_t out[8], const ge& p) {
    const fe sqrtm1 = fe_const_sqrtm1();
    const fe u1 = fe_mul(fe_add(p.Z, p.Y), fe_sub(p.Z, p.Y));
    const fe u2 = fe_mul(p.X, p.Y);
    fe invsqrt;
    fe_sqrt_ratio_m1(invsqrt, fe_one(), fe_mul(u1, fe_sq(u2)));
    const fe den1 = fe_mul(invsqrt, u1);
    const fe den2 = fe_mul(invsqrt, u2);
    const fe z_inv = fe_mul(fe_mul(den1, den2), p.T);
    const fe ix = fe_mul(p.X, sqrtm1);
    const fe iy = fe_mul(p.Y, sqrtm1);
    const fe ench = fe_mul(den1, fe_const_invsqrt_a_minus_d());
    const bool rotate = fe_isneg(fe_mul(p.T, z_inv));
    const fe x = fe_select(rotate, iy, p.X);
    fe y = fe_select(rotate, ix, p.Y);
    const fe den_inv = fe_select(rotate, ench, den2);
    const bool negy = fe_isneg(fe_mul(x, z_inv));
    y = fe_select(negy, fe_carry(fe_neg(y)), y);
    const fe s = fe_abs(fe_mul(den_inv, fe_sub(p.Z, y)));
    fe_towords(out, s);
}

// RFC 9496 4.3.4 MAP + from_uniform_bytes (generator derivation; runs on the host at init)
ZKP_HD inline ge ge_elligator_map(const fe& t) {
    const fe one = fe_one(), d = fe_const_d(), sqrtm1 = fe_const_sqrtm1();
    const fe r = fe_mul(sqrtm1, fe_sq(t));
    const fe u = fe_mul(fe_add(r, one), fe_const_one_minus_d_sq());
    const fe rd = fe_mul(r, d);
    const fe v = fe_mul(fe_carry(fe_sub(fe_neg(one), rd)), fe_add(r, d));
    fe s;
    const bool was_square = fe_sqrt_ratio_m1(s, u, v);
    const fe s_prime = fe_carry(fe_neg(fe_abs(fe_mul(s, t)) ));
    s = fe_select(was_square, fe_carry(s), s_prime);
    const fe c = fe_select(was_square, fe_carry(fe_neg(one)), r);
    const fe N = fe_carry(fe_sub(fe_mul(fe_mul(c, fe_carry(fe_sub(r, one))), fe_const_d_minus_one_sq()), v));
    const fe sv = fe_mul(s, v);
    const fe w0 = fe_add(sv, sv);
    const fe w1 = fe_mul(N, fe_const_sqrt_ad_minus_one());
    const fe ss = fe_sq(s);
    const fe w2 = fe_carry(fe_sub(one, ss));
    const fe w3 = fe_add(one, ss);
    ge p;
    p.X = fe_mul(w0, w3);
    p.Y = fe_mul(w2, w1);
    p.Z = fe_mul(w1, w3);
    p.T = fe_mul(w0, w2);
    return p;
}

ZKP_HD inline ge ge_from_uniform_words(const uint32_t w[16]) {
    return ge_add(ge_elligator_map(fe_fromwords(w)), ge_elligator_map(fe_fromwords(w + 8)));
}

}  // namespace zkp
