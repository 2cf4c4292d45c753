// TEST INFRASTRUCTURE: host build of the BN254 device headers for the no-GPU test tier.
#include "../../libzkp_amd/csrc/bn254_g.h"
#include <string.h>
using namespace zkp;

template <class P> static void fp_op(int op, const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
    Fp<P> x = fp_from_raw<P>(a), y = fp_from_raw<P>(b), r;
    switch (op) { case 0: r = fp_mul(x, y); break; case 1: r = fp_add(x, y); break; case 2: r = fp_sub(x, y); break; case 3: r = fp_neg(x); break;
                  case 4: r = fp_inv(x); break; default: r = fp_add(fp_add(x, y), fp_sub(y, x)); r = fp_mul(r, fp_sub(x, fp_add(y, y))); }
    fp_to_raw(out, r);
}
static g1_aff load_g1(const uint32_t w[16]) { return g1_aff{fp_from_raw<FqParams>(w), fp_from_raw<FqParams>(w + 8)}; }
static g2_aff load_g2(const uint32_t w[32]) {
    return g2_aff{fq2{fp_from_raw<FqParams>(w), fp_from_raw<FqParams>(w + 8)}, fq2{fp_from_raw<FqParams>(w + 16), fp_from_raw<FqParams>(w + 24)}};
}
extern "C" {
void emul_fq_op(int op, const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) { fp_op<FqParams>(op, a, b, out); }
void emul_fr_op(int op, const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) { fp_op<FrParams>(op, a, b, out); }
void emul_fr_from_wide(const uint32_t w[16], uint32_t out[8]) { fp_to_raw(out, fp_from_wide<FrParams>(w)); }
// serialize(k1*P + k2*Q) with P, Q affine (raw coords); uses madd for the first term path, add, dbl via jac_mul_raw
void emul_g1_lincomb(const uint32_t p[16], const uint32_t q[16], const uint32_t k1[8], const uint32_t k2[8], uint32_t out[16]) {
    g1_jac a = jac_mul_raw(jac_from_aff(load_g1(p)), k1), b = jac_mul_raw(jac_from_aff(load_g1(q)), k2);
    g1_serialize(out, jac_add(a, b));
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
}
