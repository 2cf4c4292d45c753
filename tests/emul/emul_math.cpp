// TEST INFRASTRUCTURE: compiles the device math headers for the host so that the no-GPU test tier can
// check them against the oracle.  Not part of the product library.
#include "../../libzkp_amd/csrc/fe25519.h"
#include "../../libzkp_amd/csrc/sc25519.h"
#include "../../libzkp_amd/csrc/ge25519.h"
#include "../../libzkp_amd/csrc/keccak.h"
#include <string.h>
using namespace zkp;

extern "C" {
// op: 0 mul, 1 sq, 2 add, 3 sub(a-b), 4 neg, 5 pow22523, 6 abs
void emul_fe_op(int op, const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
    fe x = fe_fromwords(a), y = fe_fromwords(b), r;
    switch (op) {
        case 0: r = fe_mul(x, y); break;
        case 1: r = fe_sq(x); break;
        case 2: r = fe_add(x, y); break;
        case 3: r = fe_sub(x, y); break;
        case 4: r = fe_neg(x); break;
        case 5: r = fe_pow22523(x); break;
        default: r = fe_abs(x); break;
    }
    fe_towords(out, r);
}
// loose-input stress: computes (a+b+2p-c) * (a-b+2p) style products to exercise bounds
void emul_fe_loose(const uint32_t a[8], const uint32_t b[8], const uint32_t c[8], uint32_t out[8]) {
    fe x = fe_fromwords(a), y = fe_fromwords(b), z = fe_fromwords(c);
    fe f = fe_sub(fe_add(x, x), z);   // loose (< 2^28)
    fe g = fe_sub(y, z);              // < 1.5*2^27
    fe_towords(out, fe_mul(f, g));
}
int emul_sqrt_ratio(const uint32_t u[8], const uint32_t v[8], uint32_t out[8]) {
    fe r; bool sq = fe_sqrt_ratio_m1(r, fe_fromwords(u), fe_fromwords(v)); fe_towords(out, r); return sq;
}
// scalar ops on raw canonical words: 0 mul, 1 add, 2 sub, 3 invert (safegcd), 4 neg, 5 invert (Fermat)
void emul_sc_op(int op, const uint32_t a[8], const uint32_t b[8], uint32_t out[8]) {
    sc x, y; memcpy(x.v, a, 32); memcpy(y.v, b, 32);
    x = sc_from_raw256(x); y = sc_from_raw256(y);
    sc r;
    switch (op) { case 0: r = sc_mul(x, y); break; case 1: r = sc_add(x, y); break; case 2: r = sc_sub(x, y); break; case 3: r = sc_invert(x); break; case 5: r = sc_invert_fermat(x); break; default: r = sc_neg(x); }
    r = sc_to_raw(r); memcpy(out, r.v, 32);
}
void emul_sc_from_wide(const uint32_t w[16], uint32_t out[8]) { sc r = sc_to_raw(sc_from_wide(w)); memcpy(out, r.v, 32); }
void emul_sc_recode1024(const uint32_t a[8], int16_t digits[26]) {
    sc x; memcpy(x.v, a, 32); uint32_t p[13]; sc_recode_signed1024(p, x); memcpy(digits, p, 52);
}
void emul_tape_draw64(const uint32_t seed[8], uint32_t pidx, uint32_t slot, uint32_t out[16]) { tape_draw64(out, seed, pidx, slot); }
void emul_from_uniform_encode(const uint32_t w[16], uint32_t out[8]) { ge p = ge_from_uniform_words(w); ge_ristretto_encode(out, p); }
// encode(a*P + b*Q) where P, Q = from_uniform(w1), from_uniform(w2); a, b small ints via repeated add/dbl
void emul_ge_lincomb(const uint32_t w1[16], const uint32_t w2[16], uint32_t a, uint32_t b, uint32_t out[8]) {
    ge P = ge_from_uniform_words(w1), Q = ge_from_uniform_words(w2), acc = ge_identity();
    for (int i = 31; i >= 0; i--) { acc = ge_dbl(acc); if ((a >> i) & 1) acc = ge_add(acc, P); if ((b >> i) & 1) acc = ge_add(acc, Q); }
    ge_ristretto_encode(out, acc);
}
// Merlin KAT style: Transcript::new(label); append(l1, m1); challenge(l2, n words)
void emul_merlin_kat(const char* label, const char* l1, const char* m1, const char* l2, uint32_t nwords, uint32_t* out) {
    uint32_t st[50]; Strobe s; s.base = st; s.stride = 1;
    merlin_init(s, label, (uint32_t)strlen(label));
    merlin_append_bytes(s, l1, (uint32_t)strlen(l1), m1, (uint32_t)strlen(m1));
    merlin_challenge_words(s, l2, (uint32_t)strlen(l2), out, nwords);
}
}
