// One pairing check for a whole batch of Groth16 envelopes (round 4).
//
// Per envelope j the check of SnarkBackend::verify_*_zk (/root/reference/src/backend/snark.rs:377-401,455-495 -> ark-groth16) is
//     e(A_j, B_j) e(-L_j, gamma) e(-C_j, delta) e(-alpha, beta) == 1,        L_j = sum_t x_{j,t} IC_t.
// With fresh 128-bit weights rho_j from the OS, all of them hold (up to 2^-128: the values live in the order-r target group once every B_j
// has passed its subgroup check) iff
//     prod_j e(rho_j A_j, B_j)  *  e(-sum_j rho_j L_j, gamma)  *  e(-sum_j rho_j C_j, delta)  *  e(-(sum_j rho_j) alpha, beta) == 1.
// What is left per envelope is ONE Miller loop -- (B_j, rho_j A_j): chain A of the Fq2 machine -- the subgroup check of B_j and two short
// scalar multiplications in G1 (g1_mul_weight); the Miller loops on gamma and delta and the final exponentiation (57 % of the machine's work per envelope)
// happen once per batch, for a "virtual envelope" V that goes through the same chains as any other:
//     A_V = (1 - s) alpha,  B_V = beta,  L_V = sum_t S_t IC_t  with  S_t = sum_j rho_j x_{j,t},  C_V = sum_j rho_j C_j,  s = S_0 = sum_j rho_j
// (the finishing chain multiplies the key's constant e(-alpha, beta) in, hence 1 - s), and whose chain-A value is multiplied by the
// product of the batch's chain-A values before the finishing chain runs.  sum_j rho_j L_j needs no curve arithmetic per envelope: L_j is
// linear in the key's fixed points, so the S_t are sums in the scalar field and L_V is one walk of the key's window tables.
// A batch that fails (or holds an envelope the machine leaves to the lane-per-chain path) is verified again envelope by envelope: the
// verdicts are always those of the per-envelope check.  upstream has no batch verification; this is an internal fast path like the range
// verifier's (bp_verify.h RlcView).
#pragma once
#include "g16_verify.h"

namespace zkp {

// Weights.  An envelope's 16 random bytes are two 64-bit halves (rho1, rho2) and its weight is w = rho1 + lambda rho2 mod r, lambda the
// eigenvalue of the endomorphism phi(x, y) = (beta x, y) (bn254_g.h: fr_glv_split, fq_glv_beta): the 2^128 pairs give 2^128 distinct
// weights (a collision would be a vector of the GLV lattice inside the 2^64 box; its shortest vectors have norm ~2^127), which is all the
// soundness argument needs, and w P = rho1 P + rho2 phi(P) is ONE 64-step joint double-and-add over P, phi(P) and P + phi(P) = -phi^2(P)
// = (-(beta + 1) x, -y) -- three affine points that cost one field multiplication -- instead of a 128-step ladder.
ZKP_HD inline fr g16_rlc_lambda() {
    const uint32_t w[8] = {0xb99c90ddu, 0x8b17ea66u, 0x8d8daaa7u, 0x5bfc4108u, 0x41a91758u, 0xb3c4d79du, 0x00000000u, 0x00000000u};
    return fp_from_raw<FrParams>(w);
}
ZKP_HD inline fr g16_rlc_weight(const uint32_t rho[4]) {
    const uint32_t w1[8] = {rho[0], rho[1], 0, 0, 0, 0, 0, 0}, w2[8] = {rho[2], rho[3], 0, 0, 0, 0, 0, 0};
    return fp_add(fp_from_raw<FrParams>(w1), fp_mul(g16_rlc_lambda(), fp_from_raw<FrParams>(w2)));
}
ZKP_HD_NOINLINE inline g1_jac g1_mul_weight(const g1_aff& p, const uint32_t rho[4]) {
    const fq bx = fq_mul(fq_glv_beta(), p.x);
    const g1_aff t2{bx, p.y}, t3{fq_neg(fq_add(bx, p.x)), fq_neg(p.y)};
    const uint64_t r1 = (uint64_t)rho[0] | ((uint64_t)rho[1] << 32), r2 = (uint64_t)rho[2] | ((uint64_t)rho[3] << 32);
    g1_jac acc = jac_infinity<fq>();
    for (int i = 63; i >= 0; i--) {
        acc = jac_dbl(acc);
        const uint32_t sel = (uint32_t)((r1 >> i) & 1u) | ((uint32_t)((r2 >> i) & 1u) << 1);
        if (sel == 0) continue;
        g1_aff q;
        ZKP_UNROLL for (int k = 0; k < 10; k++) { q.x.v[k] = sel == 1 ? p.x.v[k] : sel == 2 ? t2.x.v[k] : t3.x.v[k]; q.y.v[k] = sel == 3 ? t3.y.v[k] : p.y.v[k]; }
        acc = jac_madd(acc, q);
    }
    return acc;
}
// number of public-input terms of a circuit (= gamma_abc_g1 length) and rho_j x_{j,t} (Montgomery form); false: the term is zero
ZKP_HD inline bool g16_rlc_term(const G16Inputs& h, uint32_t t, const fr& rho, fr& out) {
    if (t == 0) { out = rho; return true; }
    if (t == 1) { out = fp_mul(rho, fp_from_raw<FrParams>(h.c)); return true; }
    const uint32_t i = (t - 2) % G16_MAX_SET;
    if (i >= h.n) return false;
    if (t >= 2 + G16_MAX_SET) { out = rho; return true; }                // is_real = 1
    const uint64_t v = g16_set_element(h, i);
    if (v == 0) return false;
    out = fp_mul(rho, fp_from_u64<FrParams>(v));
    return true;
}
ZKP_HD inline fr g16_rlc_one_minus(const fr& s0) { return fp_sub(fp_one<FrParams>(), s0); }          // the scalar of alpha (point n_ic of the key's window tables)

// Miller value in the machine's slots (coefficient k of w^k in slot SLOT_F0 + k: tools/gen_fq2vm.py coeffs()) <-> bn254_pairing.h's tower
ZKP_HD inline fq12 fq12_from_coeffs(const fq2 c[6]) { return fq12{fq6{c[0], c[2], c[4]}, fq6{c[1], c[3], c[5]}}; }
ZKP_HD inline void fq12_to_coeffs(fq2 c[6], const fq12& f) { c[0] = f.c0.a0; c[2] = f.c0.a1; c[4] = f.c0.a2; c[1] = f.c1.a0; c[3] = f.c1.a1; c[5] = f.c1.a2; }

// The whole batch check on one thread, through the lane-per-chain pairing code (host emulation: tests/emul; documents what the kernels of
// fq2vm_kernels.hip compute between them).  envs[j] / lens[j]: the envelopes; rho: 4 words each.  Returns 1 accepted, 0 rejected,
// 2 the batch holds an envelope that the fast path leaves to the per-envelope check (refused header / point, point at infinity, B outside the subgroup).
ZKP_HD inline int g16_rlc_check(int kind, const G16Vk& vk, const g1_aff& alpha, const g2_aff& beta, uint32_t count, const uint8_t* const* envs, const uint32_t* lens, const uint32_t* rho) {
    fq12 prod = fq12_one();
    g1_jac Cv = jac_infinity<fq>();
    fr S[2 + 2 * G16_MAX_SET];
    for (uint32_t t = 0; t < vk.n_ic; t++) S[t] = fp_zero<FrParams>();
    for (uint32_t j = 0; j < count; j++) {
        G16Inputs h;
        if (!g16_header(kind, vk, envs[j], lens[j], h)) return 2;
        g1_aff A, C; g2_aff B;
        if (g1_from_ark(A, h.proof) != 1 || g2_from_ark(B, h.proof + 64) != 1 || g1_from_ark(C, h.proof + 192) != 1) return 2;
        if (!g2_in_subgroup(B)) return 2;
        const fr w = g16_rlc_weight(rho + 4 * j);
        for (uint32_t t = 0; t < vk.n_ic; t++) { fr x; if (g16_rlc_term(h, t, w, x)) S[t] = fp_add(S[t], x); }
        g1_aff rA;
        if (!jac_to_aff(rA, g1_mul_weight(A, rho + 4 * j))) return 2;
        prod = fq12_mul(prod, miller_loop(B, rA));
        Cv = jac_add(Cv, g1_mul_weight(C, rho + 4 * j));
    }
    g1_jac Lv = jac_infinity<fq>();
    for (uint32_t t = 0; t < vk.n_ic; t++) { uint32_t k[8]; fp_to_raw(k, S[t]); Lv = jac_add(Lv, jac_mul_raw(jac_from_aff(ld_ic(vk, t)), k)); }
    uint32_t k1[8]; fp_to_raw(k1, g16_rlc_one_minus(S[0]));
    g1_aff Av, La, Ca;
    if (!jac_to_aff(Av, jac_mul_raw(jac_from_aff(alpha), k1)) || !jac_to_aff(La, Lv) || !jac_to_aff(Ca, Cv)) return 2;
    prod = fq12_mul(prod, miller_loop(beta, Av));
    return g16_finish(vk, prod, miller_loop(vk.gamma, aff_neg(La)), miller_loop(vk.delta, aff_neg(Ca))) ? 1 : 0;
}

}  // namespace zkp
