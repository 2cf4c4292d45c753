// TEST INFRASTRUCTURE: host build of the BN254 device headers for the no-GPU test tier.
#include "../../libzkp_amd/csrc/bn254_pairing.h"
#include "../../libzkp_amd/csrc/bn254_fr9.h"
#include <string.h>
using namespace zkp;

template <class P> static void fp_op(int op, const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
    Fp<P> x = fp_from_raw<P>(a), y = fp_from_raw<P>(b), r;
    switch (op) { case 0: r = fp_mul(x, y); break; case 1: r = fp_add(x, y); break; case 2: r = fp_sub(x, y); break; case 3: r = fp_neg(x); break;
                  case 4: r = fp_inv(x); break; default: r = fp_add(fp_add(x, y), fp_sub(y, x)); r = fp_mul(r, fp_sub(x, fp_add(y, y))); }
    fp_to_raw(out, r);
}
static void fq_op(int op, const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
    fq x = fq_from_raw(a), y = fq_from_raw(b), r;
    switch (op) { case 0: r = fq_mul(x, y); break; case 1: r = fq_add(x, y); break; case 2: r = fq_sub(x, y); break; case 3: r = fq_neg(x); break;
                  case 4: r = fq_inv(x); break; default: r = fq_add(fq_add(x, y), fq_sub(y, x)); r = fq_mul(r, fq_sub(x, fq_add(y, y))); }
    fq_to_raw(out, r);
}
static g1_aff load_g1(const uint32_t w[16]) { return g1_aff{fq_from_raw(w), fq_from_raw(w + 8)}; }
static g2_aff load_g2(const uint32_t w[32]) {
    return g2_aff{fq2{fq_from_raw(w), fq_from_raw(w + 8)}, fq2{fq_from_raw(w + 16), fq_from_raw(w + 24)}};
}
extern "C" {
void emul_fq_op(int op, const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) { fq_op(op, a, b, out); }
// lazy-limb forms on loose inputs: x' = x + ka*p (limb-wise, uncarried), y' = y + kb*p; op 0: mul, 1: (x'+y') * (x' + 4p - y'), 2: reduce_weak(x' + 8p - y'),
// 3: to_raw(x') directly, 4: sq(x' + 16p - 2^2 y) with y canonical
void emul_fq_lazy(int op, const uint32_t a[8], const uint32_t b[8], int ka, int kb, uint32_t out[8]) {
    fq x = fq_from_raw(a), y = fq_from_raw(b), pl, r;
    for (int i = 0; i < 10; i++) pl.v[i] = fq_pl(i);
    const fq y0 = y;
    for (int i = 0; i < ka; i++) x = fq_add_l(x, pl);
    for (int i = 0; i < kb; i++) y = fq_add_l(y, pl);
    switch (op) { case 0: r = fq_mul(x, y); break; case 1: r = fq_mul(fq_add_l(x, y), fq_sub_k4(x, fq_reduce_weak(y))); break;
                  case 2: r = fq_reduce_weak(fq_sub_k8(x, fq_carry(y))); break; case 3: r = x; break;
                  default: r = fq_sq(fq_sub_k16(x, fq_dbl_l(fq_dbl_l(y0)))); }
    fq_to_raw(out, r);
}
// the MSM inner loop as the kernels run it: acc = O; acc += sign_i * T_i with the lazy mixed addition; out = serialize(acc)
void emul_g1_lazy_chain(const uint32_t o[16], const uint32_t* pts, const int* signs, int n, uint32_t out[16]) {
    g1_jac acc = jac_dbl(jac_from_aff(load_g1(o)));                 // Z != 1
    for (int i = 0; i < n; i++) {
        g1_aff q = load_g1(pts + 16 * i);
        q.y = fq_select(signs[i] < 0, fq_sub_k4(fq_zero(), q.y), q.y);
        acc = g1_madd_lazy(acc, q);
    }
    g1_serialize(out, acc);
}
// the same chain in the XYZZ coordinates of k_msm_gather<G1Msm>: Jacobian -> XYZZ -> additions -> Jacobian
void emul_g1_xyzz_chain(const uint32_t o[16], const uint32_t* pts, const int* signs, int n, uint32_t out[16]) {
    g1_xyzz acc = xyzz_from_jac(jac_dbl(jac_from_aff(load_g1(o))));
    for (int i = 0; i < n; i++) {
        g1_aff q = load_g1(pts + 16 * i);
        q.y = fq_select(signs[i] < 0, fq_sub_k4(fq_zero(), q.y), q.y);
        acc = g1_mmadd_lazy(acc, q);
    }
    g1_serialize(out, jac_from_xyzz(acc));
}
// the chain on nine 29-bit limbs (g1_mmadd9, the form the kernel actually runs): table entries converted as k_g16_build_table
// converts them; returns the largest limb seen in any accumulator coordinate's low limbs (must stay below 2^29) through *max_limb
void emul_g1_xyzz9_chain(const uint32_t o[16], const uint32_t* pts, const int* signs, int n, uint32_t out[16], uint32_t* max_limb) {
    g1_xyzz9 acc = xyzz9_from_jac(jac_dbl(jac_from_aff(load_g1(o))));
    uint32_t mx = 0;
    for (int i = 0; i < n; i++) {
        const g1_aff a = load_g1(pts + 16 * i);
        uint32_t e[16]; fq9_pack8(e, fq9_from_fq(a.x)); fq9_pack8(e + 8, fq9_from_fq(a.y));      // the 64-byte table entry
        g1_aff9 q{fq9_unpack8(e), fq9_unpack8(e + 8)};
        q.y = fq9_select(signs[i] < 0, fq9_neg_k<4>(q.y), q.y);
        acc = g1_mmadd9(acc, q);
        for (int k = 0; k < 9; k++) { const uint32_t w[4] = {acc.X.v[k], acc.Y.v[k], acc.ZZ.v[k], acc.ZZZ.v[k]}; for (uint32_t x : w) if (x > mx) mx = x; }
    }
    *max_limb = mx;
    g1_serialize(out, jac_from_xyzz9(acc));
}
// fq <-> fq9 round trip and one product through the nine-limb form: out = a * b mod p (raw in, raw out)
void emul_fq9_mul(const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
    const fq9 x = fq9_from_fq(fq_from_raw(a)), y = fq9_from_fq(fq_from_raw(b));
    fq_to_raw(out, fq9_to_fq(fq9_mul(x, y)));
}
// the key-table form of a coordinate: eight words in, the nine limbs the loop sees out, and those limbs packed again
void emul_fq9_pack(const uint32_t w[8], uint32_t limbs[9], uint32_t again[8]) {
    const fq9 v = fq9_unpack8(w);
    for (int i = 0; i < 9; i++) limbs[i] = v.v[i];
    fq9_pack8(again, v);
}
// what k_g16_build_table stores for the coordinate a (raw canonical in): pack8(fq9_from_fq(to Montgomery form)), and the nine limbs before packing
void emul_fq9_entry(const uint32_t a[8], uint32_t packed[8], uint32_t limbs[9]) {
    const fq9 v = fq9_from_fq(fq_from_raw(a));
    for (int i = 0; i < 9; i++) limbs[i] = v.v[i];
    fq9_pack8(packed, v);
}
void emul_fq9_ops(const uint32_t a[8], const uint32_t b[8], uint32_t sq[8], uint32_t fused[8], uint32_t sub[8]) {
    const fq9 x = fq9_from_fq(fq_from_raw(a)), y = fq9_from_fq(fq_from_raw(b));
    fq_to_raw(sq, fq9_to_fq(fq9_sq(x)));                                        // a^2
    fq_to_raw(fused, fq9_to_fq(fq9_mul_add2(x, x, fq9_neg_k<8>(x), y)));        // a^2 + (8p - a) b
    const fq9 one = fq9_from_fq(fq_one());                                      // products with it bring a value below 1.05 p
    fq_to_raw(sub, fq9_to_fq(fq9_sub2_k4(x, fq9_mul(y, one), fq9_mul(fq9_sub_k<4>(y, x), one))));      // a - b - 2 (b - a) = 3 (a - b); subtrahends < 3.2 p
}
void emul_g2_lazy_chain(const uint32_t o[32], const uint32_t* pts, const int* signs, int n, uint32_t out[32]) {
    g2_jac acc = jac_dbl(jac_from_aff(load_g2(o)));
    for (int i = 0; i < n; i++) {
        g2_aff q = load_g2(pts + 32 * i);
        acc = g2_madd_lazy(acc, q, signs[i] < 0);
    }
    g2_serialize(out, acc);
}
// the G2 chain on nine 29-bit limbs (g2_mmadd9), entries converted as the table builder converts them
void emul_g2_xyzz9_chain(const uint32_t o[32], const uint32_t* pts, const int* signs, int n, uint32_t out[32], uint32_t* max_limb) {
    g2_xyzz9 acc = g2_xyzz9_from_jac(jac_dbl(jac_from_aff(load_g2(o))));
    uint32_t mx = 0;
    for (int i = 0; i < n; i++) {
        const g2_aff a = load_g2(pts + 32 * i);
        uint32_t e[32];                                                             // the 128-byte table entry
        fq9_pack8(e, fq9_from_fq(a.x.c0)); fq9_pack8(e + 8, fq9_from_fq(a.x.c1)); fq9_pack8(e + 16, fq9_from_fq(a.y.c0)); fq9_pack8(e + 24, fq9_from_fq(a.y.c1));
        const g2_aff9 q{fq2_9{fq9_unpack8(e), fq9_unpack8(e + 8)}, fq2_9{fq9_unpack8(e + 16), fq9_unpack8(e + 24)}};
        acc = g2_mmadd9(acc, q, signs[i] < 0);
        const fq9* c = reinterpret_cast<const fq9*>(&acc);
        for (int t = 0; t < 8; t++) for (int k = 0; k < 9; k++) if (c[t].v[k] > mx) mx = c[t].v[k];
    }
    *max_limb = mx;
    g2_serialize(out, jac_from_g2_xyzz9(acc));
}
// Fr on nine 29-bit limbs (bn254_fr9.h) through its conversions: out0 = a b, out1 = a - b (fr9_sub_k<2> on products < 2r), out2 = the
// weak reduction of a + 20 b (raw in, raw out)
void emul_fr9_ops(const uint32_t a[8], const uint32_t b[8], uint32_t mul[8], uint32_t sub[8], uint32_t red[8]) {
    const fr x = fp_from_raw<FrParams>(a), y = fp_from_raw<FrParams>(b);
    const fr9 x9 = fr9_from_fr(x), y9 = fr9_from_fr(y);
    fp_to_raw(mul, fr9_to_fr<FrParams>(fr9_mul(x9, y9)));
    fp_to_raw(sub, fr9_to_fr<FrParams>(fr9_sub_k<2>(x9, y9)));
    fr9 acc = x9; for (int i = 0; i < 20; i++) acc = fr9_add(acc, y9);          // < 30 r, unreduced
    const fr9 w = fr9_reduce_weak(acc);
    uint32_t top = 0; for (int i = 0; i < 9; i++) top |= (i < 8 && w.v[i] >> 29) ? 1u : 0u;
    fp_to_raw(red, fr9_to_fr<FrParams>(w));
    red[7] |= top << 31;                                                        // a limb of the reduced value out of range would show here
}
void emul_fr_op(int op, const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) { fp_op<FrParams>(op, a, b, out); }
void emul_fr_from_wide(const uint32_t w[16], uint32_t out[8]) { fp_to_raw(out, fp_from_wide<FrParams>(w)); }
// serialize(k1*P + k2*Q) with P, Q affine (raw coords); uses madd for the first term path, add, dbl via jac_mul_raw
void emul_g1_lincomb(const uint32_t p[16], const uint32_t q[16], const uint32_t k1[8], const uint32_t k2[8], uint32_t out[16]) {
    g1_jac a = jac_mul_raw(jac_from_aff(load_g1(p)), k1), b = jac_mul_raw(jac_from_aff(load_g1(q)), k2);
    g1_serialize(out, jac_add(a, b));
}
// k * P through the GLV split of the Groth16 C element (bn254_g.h): halves (magnitude words + sign) out, and serialize(k1 P' + k2 phi(P'))
// for P' = 2 P brought in with Z != 1; must equal serialize(2 k P)
void emul_g1_glv_mul(const uint32_t p[16], const uint32_t k[8], uint32_t halves[10], uint32_t out[16]) {
    glv_half h1, h2; fr_glv_split(k, h1, h2);
    for (int i = 0; i < 4; i++) { halves[i] = h1.mag[i]; halves[5 + i] = h2.mag[i]; }
    halves[4] = h1.neg; halves[9] = h2.neg;
    g1_jac P = jac_dbl(jac_from_aff(load_g1(p))), Q = P;
    Q.X = fq_mul(Q.X, fq_glv_beta());
    g1_serialize(out, jac_add(jac_mul_u128_signed(P, h1.mag, h1.neg), jac_mul_u128_signed(Q, h2.mag, h2.neg)));
}
void emul_g2_lincomb(const uint32_t p[32], const uint32_t q[32], const uint32_t k1[8], const uint32_t k2[8], uint32_t out[32]) {
    g2_jac a = jac_mul_raw(jac_from_aff(load_g2(p)), k1), b = jac_mul_raw(jac_from_aff(load_g2(q)), k2);
    g2_serialize(out, jac_add(a, b));
}
// sum_{i<n} sign_i * P (mixed additions incl. the exceptional cases): result = (sum signs) * P
void emul_g1_madd_chain(const uint32_t p[16], const int* signs, int n, uint32_t out[16]) {
    g1_aff P = load_g1(p); g1_jac acc = jac_infinity<fq>();
    for (int i = 0; i < n; i++) acc = jac_madd(acc, signs[i] < 0 ? aff_neg(P) : P);
    g1_serialize(out, acc);
}
void emul_g2_madd_chain(const uint32_t p[32], const int* signs, int n, uint32_t out[32]) {
    g2_aff P = load_g2(p); g2_jac acc = jac_infinity<fq2>();
    for (int i = 0; i < n; i++) acc = jac_madd(acc, signs[i] < 0 ? aff_neg(P) : P);
    g2_serialize(out, acc);
}

// Fq12 product; operands and result as 12 canonical coefficients of oracle/py/bn254.py's basis Fq[w]/(w^12 - 18 w^6 + 82)
static fq12 f12_from_poly(const uint32_t* c) {
    fq2 k[6];
    for (int i = 0; i < 6; i++) {
        const fq lo = fq_from_raw(c + 8 * i), hi = fq_from_raw(c + 8 * (i + 6));
        fq nine_hi = hi; for (int t = 0; t < 3; t++) nine_hi = fq_dbl(nine_hi); nine_hi = fq_add(nine_hi, hi);
        k[i] = fq2{fq_add(lo, nine_hi), hi};
    }
    return fq12{fq6{k[0], k[2], k[4]}, fq6{k[1], k[3], k[5]}};
}
static void f12_to_poly(uint32_t* c, const fq12& a) {
    const fq2 k[6] = {a.c0.a0, a.c1.a0, a.c0.a1, a.c1.a1, a.c0.a2, a.c1.a2};
    for (int i = 0; i < 6; i++) {
        fq nine = k[i].c1; for (int t = 0; t < 3; t++) nine = fq_dbl(nine); nine = fq_add(nine, k[i].c1);
        fq_to_raw(c + 8 * i, fq_sub(k[i].c0, nine)); fq_to_raw(c + 8 * (i + 6), k[i].c1);
    }
}
void emul_f12_mul(const uint32_t a[96], const uint32_t b[96], int square, uint32_t out[96]) {
    const fq12 x = f12_from_poly(a), y = f12_from_poly(b);
    f12_to_poly(out, square ? fq12_sq(x) : fq12_mul(x, y));
}
// prod_i ate(Q_i, P_i) == 1 ?  points as raw affine coordinates (16 / 32 words each)
// final exponentiation two ways (easy/hard split vs one 2790-bit power) and the Fq12 inverse, in the oracle's basis
void emul_f12_final_exp(const uint32_t a[96], int naive, uint32_t out[96]) { f12_to_poly(out, naive ? final_exponentiation_naive(f12_from_poly(a)) : final_exponentiation(f12_from_poly(a))); }
void emul_f12_final_exp_chain(const uint32_t a[96], uint32_t out[96]) { f12_to_poly(out, final_exponentiation_chain(f12_from_poly(a))); }
void emul_f12_frob(const uint32_t a[96], int j, uint32_t out[96]) { f12_to_poly(out, j == 2 ? fq12_frob_p2(f12_from_poly(a)) : fq12_frob_odd(f12_from_poly(a), j)); }
// squaring of the easy part's output (an element of the cyclotomic subgroup): general (0) or Granger-Scott (1)
void emul_f12_cyclo_sq(const uint32_t a[96], int gs, uint32_t out[96]) {
    const fq12 f = f12_from_poly(a);
    const fq12 e1 = fq12_mul(fq12_conj(f), fq12_inv(f)), r = fq12_mul(fq12_frob_p2(e1), e1);
    f12_to_poly(out, gs ? fq12_cyclo_sq(r) : fq12_sq(r));
}
// f * line(A, B, C) by the sparse product (1) or the general one on the embedded line (0); abc = 3 x (c0, c1) raw words
void emul_f12_mul_line(const uint32_t a[96], const uint32_t abc[48], int sparse, uint32_t out[96]) {
    auto f2 = [&](int k) { return fq2{fq_from_raw(abc + 16 * k), fq_from_raw(abc + 16 * k + 8)}; };
    const fq12 f = f12_from_poly(a);
    const fq12_line l{f2(0), f2(1), f2(2)};
    f12_to_poly(out, sparse ? fq12_mul_line(f, l) : fq12_mul(f, fq12_from_line(l.A, l.B, l.C)));
}
void emul_f12_inv(const uint32_t a[96], uint32_t out[96]) { f12_to_poly(out, fq12_inv(f12_from_poly(a))); }
int emul_pairing_product_is_one(int n, const uint32_t* g1s, const uint32_t* g2s) {
    fq12 f = fq12_one();
    for (int i = 0; i < n; i++) f = fq12_mul(f, miller_loop(load_g2(g2s + 32 * i), load_g1(g1s + 16 * i)));
    return fq12_is_one(final_exponentiation(f)) ? 1 : 0;
}
}
