// GF(2^255-19) for gfx950 VALU: ten unsigned 25.5-bit limbs in 32-bit registers (even limbs 26 bits,
// odd limbs 25 bits), 64-bit column sums via v_mad_u64_u32.  Chosen by measurement on MI355X
// (tools/fe_microbench.hip: 257 G mul/s vs 162 for 8x32 saturated limbs and 184 for 12xf64 FMA limbs).
//
// Replaces curve25519-dalek's FieldElement as used under /root/reference/src/backend/bulletproofs.rs:4-5.
//
// Limb-bound vocabulary used in the comments below:
//   carried : even limbs < 2^26, odd limbs < 2^25 + 2^18       (output of fe_mul / fe_sq / fe_carry)
//   loose   : even limbs < 2^28, odd limbs < 2^27              (sums/differences of carried values)
// fe_mul(f, g) accepts f loose and g with even limbs < 1.5*2^27, odd limbs < 1.5*2^26 (so 19*g fits 32 bits);
// tests/test_fe_bounds.py proves by interval arithmetic that no 64-bit column sum overflows.
#pragma once
#include "zkp_common.h"

namespace zkp {

struct fe { uint32_t v[10]; };

ZKP_HD inline fe fe_zero() { fe r; ZKP_UNROLL for (int i = 0; i < 10; i++) r.v[i] = 0; return r; }
ZKP_HD inline fe fe_one() { fe r = fe_zero(); r.v[0] = 1; return r; }

ZKP_HD inline void fe_carry_wide(fe& o, uint64_t h[10]) {
    uint64_t c;
    ZKP_UNROLL for (int k = 0; k < 9; k++) {
        const int bits = (k & 1) ? 25 : 26;
        c = h[k] >> bits; h[k] &= ((1ull << bits) - 1); h[k + 1] += c;
    }
    c = h[9] >> 25; h[9] &= 0x1ffffffull;
    h[0] += c * 19;
    c = h[0] >> 26; h[0] &= 0x3ffffffull; h[1] += c;
    ZKP_UNROLL for (int k = 0; k < 10; k++) o.v[k] = (uint32_t)h[k];
}

// h = f * g.  Columns are accumulated in order with the previous column's carry as the initial addend of the
// v_mad_u64_u32 chain, so the carry pass costs a shift and a mask per limb and no separate 64-bit additions.
ZKP_HD inline fe fe_mul(const fe& f, const fe& g) {
    uint32_t g19[10], f2[10];
    ZKP_UNROLL for (int i = 0; i < 10; i++) g19[i] = 19u * g.v[i];
    ZKP_UNROLL for (int i = 1; i < 10; i += 2) f2[i] = 2u * f.v[i];
    fe o;
    uint64_t c = 0;
    ZKP_UNROLL for (int k = 0; k < 10; k++) {
        uint64_t acc = c;
        ZKP_UNROLL for (int i = 0; i < 10; i++) {
            int j = k - i;
            const bool wrap = j < 0;
            if (wrap) j += 10;
            const uint32_t fi = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
            acc += (uint64_t)fi * (wrap ? g19[j] : g.v[j]);
        }
        const int bits = (k & 1) ? 25 : 26;
        o.v[k] = (uint32_t)acc & ((1u << bits) - 1);
        c = acc >> bits;
    }
    const uint64_t t = (uint64_t)o.v[0] + c * 19;
    o.v[0] = (uint32_t)t & 0x3ffffffu;
    o.v[1] += (uint32_t)(t >> 26);
    return o;
}

// h = f^2 (55 products); f loose
ZKP_HD inline fe fe_sq(const fe& f) {
    uint32_t f19[10], f2[10], f4[10];
    ZKP_UNROLL for (int i = 0; i < 10; i++) { f19[i] = 19u * f.v[i]; f2[i] = 2u * f.v[i]; f4[i] = 4u * f.v[i]; }
    uint64_t h[10];
    ZKP_UNROLL for (int k = 0; k < 10; k++) {
        uint64_t acc = 0;
        // pairs (i, j) with i <= j, i + j == k (mod 10); multiplier = (i<j ? 2 : 1) * (odd,odd ? 2 : 1) * (wrap ? 19 : 1)
        ZKP_UNROLL for (int i = 0; i < 10; i++) {
            ZKP_UNROLL for (int j = i; j < 10; j++) {
                if ((i + j) % 10 != k) continue;
                const bool wrap = (i + j) >= 10;
                const int mult = ((i < j) ? 2 : 1) * (((i & 1) && (j & 1)) ? 2 : 1);
                const uint32_t a = mult == 4 ? f4[i] : mult == 2 ? f2[i] : f.v[i];
                acc += (uint64_t)a * (wrap ? f19[j] : f.v[j]);
            }
        }
        h[k] = acc;
    }
    fe o; fe_carry_wide(o, h); return o;
}

ZKP_HD inline fe fe_add(const fe& f, const fe& g) { fe r; ZKP_UNROLL for (int i = 0; i < 10; i++) r.v[i] = f.v[i] + g.v[i]; return r; }

// h = f - g + 2p ; g must be carried; result loose when f is carried
ZKP_HD inline fe fe_sub(const fe& f, const fe& g) {
    fe r;
    r.v[0] = f.v[0] + 0x7ffffdau - g.v[0];
    ZKP_UNROLL for (int i = 1; i < 10; i++) r.v[i] = f.v[i] + ((i & 1) ? 0x3fffffeu : 0x7fffffeu) - g.v[i];
    return r;
}

// full carry pass from limbs < 2^32 to carried form
ZKP_HD inline fe fe_carry(const fe& f) {
    uint64_t h[10];
    ZKP_UNROLL for (int i = 0; i < 10; i++) h[i] = f.v[i];
    fe o; fe_carry_wide(o, h); return o;
}

ZKP_HD inline fe fe_neg(const fe& f) { return fe_sub(fe_zero(), f); }  // f carried

ZKP_HD inline fe fe_select(bool c, const fe& a, const fe& b) {  // c ? a : b
    fe r; ZKP_UNROLL for (int i = 0; i < 10; i++) r.v[i] = c ? a.v[i] : b.v[i]; return r;
}

// canonical little-endian bytes as eight 32-bit words
ZKP_HD inline void fe_towords(uint32_t w[8], const fe& f) {
    fe h = fe_carry(f);
    uint32_t q = (h.v[0] + 19) >> 26;
    ZKP_UNROLL for (int i = 1; i < 10; i++) q = (h.v[i] + q) >> ((i & 1) ? 25 : 26);
    h.v[0] += 19 * q;
    uint32_t c;
    ZKP_UNROLL for (int i = 0; i < 9; i++) {
        const int bits = (i & 1) ? 25 : 26;
        c = h.v[i] >> bits; h.v[i] &= (1u << bits) - 1; h.v[i + 1] += c;
    }
    h.v[9] &= 0x1ffffffu;
    // bit offsets 0,26,51,77,102,128,153,179,204,230
    w[0] = h.v[0] | (h.v[1] << 26);
    w[1] = (h.v[1] >> 6) | (h.v[2] << 19);
    w[2] = (h.v[2] >> 13) | (h.v[3] << 13);
    w[3] = (h.v[3] >> 19) | (h.v[4] << 6);
    w[4] = h.v[5] | (h.v[6] << 25);
    w[5] = (h.v[6] >> 7) | (h.v[7] << 19);
    w[6] = (h.v[7] >> 13) | (h.v[8] << 12);
    w[7] = (h.v[8] >> 20) | (h.v[9] << 6);
}

ZKP_HD inline fe fe_fromwords(const uint32_t w[8]) {  // drops bit 255
    fe h;
    h.v[0] = w[0] & 0x3ffffffu;
    h.v[1] = ((w[0] >> 26) | (w[1] << 6)) & 0x1ffffffu;
    h.v[2] = ((w[1] >> 19) | (w[2] << 13)) & 0x3ffffffu;
    h.v[3] = ((w[2] >> 13) | (w[3] << 19)) & 0x1ffffffu;
    h.v[4] = (w[3] >> 6) & 0x3ffffffu;
    h.v[5] = w[4] & 0x1ffffffu;
    h.v[6] = ((w[4] >> 25) | (w[5] << 7)) & 0x3ffffffu;
    h.v[7] = ((w[5] >> 19) | (w[6] << 13)) & 0x1ffffffu;
    h.v[8] = ((w[6] >> 12) | (w[7] << 20)) & 0x3ffffffu;
    h.v[9] = (w[7] >> 6) & 0x1ffffffu;
    return h;
}

ZKP_HD inline bool fe_isneg(const fe& f) { uint32_t w[8]; fe_towords(w, f); return w[0] & 1; }
ZKP_HD inline bool fe_iszero(const fe& f) {
    uint32_t w[8]; fe_towords(w, f);
    uint32_t r = 0; ZKP_UNROLL for (int i = 0; i < 8; i++) r |= w[i];
    return r == 0;
}
ZKP_HD inline bool fe_eq(const fe& f, const fe& g) {
    uint32_t a[8], b[8]; fe_towords(a, f); fe_towords(b, g);
    uint32_t r = 0; ZKP_UNROLL for (int i = 0; i < 8; i++) r |= a[i] ^ b[i];
    return r == 0;
}
ZKP_HD inline fe fe_abs(const fe& f) { fe c = fe_carry(f); return fe_select(fe_isneg(c), fe_neg(c), c); }

ZKP_HD inline fe fe_sqn(fe x, int n) {
    for (int i = 0; i < n; i++) x = fe_sq(x);
    return x;
}

// z^((p-5)/8) = z^(2^252-3)
ZKP_HD inline fe fe_pow22523(const fe& z) {
    fe t0 = fe_sq(z);
    fe t1 = fe_sqn(t0, 2);
    t1 = fe_mul(z, t1);
    t0 = fe_mul(t0, t1);
    t0 = fe_sq(t0);
    t0 = fe_mul(t1, t0);
    t1 = fe_sqn(t0, 5);
    t0 = fe_mul(t1, t0);
    t1 = fe_sqn(t0, 10);
    t1 = fe_mul(t1, t0);
    fe t2 = fe_sqn(t1, 20);
    t1 = fe_mul(t2, t1);
    t1 = fe_sqn(t1, 10);
    t0 = fe_mul(t1, t0);
    t1 = fe_sqn(t0, 50);
    t1 = fe_mul(t1, t0);
    t2 = fe_sqn(t1, 100);
    t1 = fe_mul(t2, t1);
    t1 = fe_sqn(t1, 50);
    t0 = fe_mul(t1, t0);
    t0 = fe_sqn(t0, 2);
    return fe_mul(t0, z);
}

// field constants (canonical little-endian words; values cross-checked against oracle/py in tests)
ZKP_HD inline fe fe_const_d() { const uint32_t w[8] = {0x135978a3u, 0x75eb4dcau, 0x4141d8abu, 0x00700a4du, 0x7779e898u, 0x8cc74079u, 0x2b6ffe73u, 0x52036ceeu}; return fe_fromwords(w); }
ZKP_HD inline fe fe_const_sqrtm1() { const uint32_t w[8] = {0x4a0ea0b0u, 0xc4ee1b27u, 0xad2fe478u, 0x2f431806u, 0x3dfbd7a7u, 0x2b4d0099u, 0x4fc1df0bu, 0x2b832480u}; return fe_fromwords(w); }

// RFC 9496 section 4.2: r = sqrt(u/v) (or sqrt(i*u/v)), returns was_square
ZKP_HD inline bool fe_sqrt_ratio_m1(fe& r, const fe& u, const fe& v) {
    const fe sqrtm1 = fe_const_sqrtm1();
    fe v3 = fe_mul(fe_sq(v), v);
    fe v7 = fe_mul(fe_sq(v3), v);
    fe t = fe_pow22523(fe_mul(u, v7));
    fe rr = fe_mul(fe_mul(u, v3), t);
    fe check = fe_mul(fe_sq(rr), v);
    fe uc = fe_carry(u);
    fe neg_u = fe_neg(uc);
    fe neg_u_i = fe_mul(neg_u, sqrtm1);
    bool correct = fe_eq(check, uc), flipped = fe_eq(check, neg_u), flipped_i = fe_eq(check, neg_u_i);
    fe r_prime = fe_mul(rr, sqrtm1);
    rr = fe_select(flipped || flipped_i, r_prime, rr);
    r = fe_abs(rr);
    return correct || flipped;
}

}  // namespace zkp
