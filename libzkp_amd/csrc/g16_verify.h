// Groth16 verification of libzkp's equality / membership envelopes:
// SnarkBackend::verify_equality_zk (/root/reference/src/backend/snark.rs:377-401) and verify_membership_zk (:455-495),
// i.e. ark-groth16's  e(A, B) == e(alpha, beta) e(sum x_i IC_i, gamma) e(C, delta)  with the public-input order of
// snark.rs:397-398,482-492.  Per envelope: one thread parses and accumulates the public inputs, three threads run the three
// Miller loops, one thread multiplies and exponentiates (pairing: bn254_pairing.h); restated in
// oracle/py/groth16.py: verify / verify_equality_with_commitment / verify_membership.
#pragma once
#include "bn254_pairing.h"
#include "g16_steps.h"

namespace zkp {

// verifying key in device memory (built by the host from the loaded proving key, which starts with the vk)
struct G16Vk {
    g2_aff gamma, delta;
    g2_aff beta;                   // B of the batch check's virtual envelope (g16_rlc.h); the window tables carry alpha as point n_ic
    fq12 ml_alpha_beta;            // Miller loop value of (beta, -alpha): the constant factor of the check
    uint32_t n_ic;                 // gamma_abc_g1 length (1 + public inputs)
    const uint32_t* ic;            // [n_ic][20] affine points, Montgomery limbs
    const uint32_t* ic_table;      // optional [n_ic][G16V_NWIN windows][G16V_NENT entries][20]: entry e of window w = (e + 1) * 2^(WBITS w) * IC_i (affine)
};

ZKP_HD inline uint32_t ld_u32_le(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
ZKP_HD inline void ld_le_words(uint32_t w[8], const uint8_t* p) {
    for (int i = 0; i < 8; i++) w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
}
ZKP_HD inline bool fq_raw_lt_p(const uint32_t w[8]) {
    for (int i = 7; i >= 0; i--) { if (w[i] < FqParams::mod(i)) return true; if (w[i] > FqParams::mod(i)) return false; }
    return false;
}
ZKP_HD inline bool fr_raw_lt_r(const uint32_t w[8]) {
    for (int i = 7; i >= 0; i--) { if (w[i] < FrParams::mod(i)) return true; if (w[i] > FrParams::mod(i)) return false; }
    return false;
}
// ark-serialize uncompressed G1: x || y little-endian, flags in the top two bits of the last byte (bn254_g.h: g1_serialize).
// Returns 0 invalid, 1 finite point, 2 point at infinity.  Exactly ark's rules (SWFlags::from_u8 + deserialize_with_mode,
// Compress::No, Validate::Yes): both flag bits set is an error; for a finite point the sign flag is NOT compared with y
// (uncompressed deserialisation takes y from the bytes and ignores that bit); coordinates must be canonical and on the curve.
ZKP_HD_NOINLINE inline int g1_from_ark(g1_aff& out, const uint8_t b[64]) {
    const uint32_t flags = b[63] & 0xC0u;
    if (flags == 0xC0u) return 0;
    if (flags & 0x40u) return 2;
    uint32_t xw[8], yw[8];
    ld_le_words(xw, b); ld_le_words(yw, b + 32); yw[7] &= 0x3FFFFFFFu;
    if (!fq_raw_lt_p(xw) || !fq_raw_lt_p(yw)) return 0;
    out.x = fq_from_raw(xw); out.y = fq_from_raw(yw);
    const fq rhs = fq_add(fq_mul(fq_sq(out.x), out.x), fq_from_u64(3));
    if (!fq_eq(fq_sq(out.y), rhs)) return 0;
    return 1;
}
// (the subgroup check of a finite point is g2_in_subgroup: the most expensive part, which the GPU path runs on a lane of its own)
ZKP_HD_NOINLINE inline int g2_from_ark(g2_aff& out, const uint8_t b[128]) {
    const uint32_t flags = b[127] & 0xC0u;
    if (flags == 0xC0u) return 0;
    if (flags & 0x40u) return 2;
    uint32_t w[4][8];
    for (int k = 0; k < 4; k++) ld_le_words(w[k], b + 32 * k);
    w[3][7] &= 0x3FFFFFFFu;
    for (int k = 0; k < 4; k++) if (!fq_raw_lt_p(w[k])) return 0;
    out.x = fq2{fq_from_raw(w[0]), fq_from_raw(w[1])}; out.y = fq2{fq_from_raw(w[2]), fq_from_raw(w[3])};
    // the twist's constant 3 / (9 + u) (oracle/py/bn254.py B2), as words: one Montgomery conversion each instead of the Fq inversion the
    // quotient cost every lane (a third of the parsing kernel's time)
    const uint32_t B2C0[8] = {0x24a138e5u, 0x3267e6dcu, 0x59dbefa3u, 0xb5b4c5e5u, 0x1be06ac3u, 0x81be1899u, 0xceb8aaaeu, 0x2b149d40u};
    const uint32_t B2C1[8] = {0x85c315d2u, 0xe4a2bd06u, 0xe52d1852u, 0xa74fa084u, 0xeed8fdf4u, 0xcd2cafadu, 0x3af0fed4u, 0x009713b0u};
    const fq2 b2{fq_from_raw(B2C0), fq_from_raw(B2C1)};
    if (!fq2_eq(f_sq(out.y), f_add(f_mul(f_sq(out.x), out.x), b2))) return 0;
    return 1;
}
// G2 has a cofactor: r * Q must be the identity
ZKP_HD_NOINLINE inline bool g2_in_subgroup(const g2_aff& q) {
    uint32_t rw[8]; for (int i = 0; i < 8; i++) rw[i] = FrParams::mod(i);
    return jac_is_inf(jac_mul_raw(jac_from_aff(q), rw));
}
// k * P for a 64-bit scalar (set elements)
ZKP_HD inline g1_jac g1_mul_u64(const g1_aff& p, uint64_t k) {
    g1_jac acc = jac_infinity<fq>();
    for (int i = 63; i >= 0; i--) {
        acc = jac_dbl(acc);
        if ((k >> i) & 1u) acc = jac_madd(acc, p);
    }
    return acc;
}
ZKP_HD inline g1_aff ld_ic(const G16Vk& vk, uint32_t i) {
    g1_aff p; const uint32_t* q = vk.ic + (size_t)i * 20;
    for (int k = 0; k < 10; k++) { p.x.v[k] = q[k]; p.y.v[k] = q[10 + k]; }
    return p;
}
// k * IC_i for a raw scalar that fits nwin windows: one mixed addition per non-zero signed radix-2^WBITS digit when the key
// carries window tables (digits are recoded on the fly, low window first), the generic ladder otherwise (host emulation of
// large keys)
ZKP_HD_NOINLINE inline g1_jac g16_ic_mul(const G16Vk& vk, uint32_t i, const uint32_t k[8], uint32_t nwin) {
    if (vk.ic_table == nullptr) return jac_mul_raw(jac_from_aff(ld_ic(vk, i)), k);
    g1_jac acc = jac_infinity<fq>();
    uint32_t carry = 0;
    for (uint32_t w = 0; w < nwin; w++) {
        const uint32_t bit = G16V_WBITS * w, wd = bit >> 5, sh = bit & 31u;
        uint32_t x = wd < 8 ? k[wd] >> sh : 0u;
        if (sh + G16V_WBITS > 32 && wd + 1 < 8) x |= k[wd + 1] << (32 - sh);
        const uint32_t dd = (x & ((1u << G16V_WBITS) - 1u)) + carry;            // 0 .. 2^WBITS
        carry = dd > G16V_NENT ? 1u : 0u;
        const int32_t d = (int32_t)dd - (int32_t)(carry << G16V_WBITS);         // [-(NENT - 1), NENT]
        if (d == 0) continue;
        const uint32_t* e = vk.ic_table + (((size_t)i * G16V_NWIN + w) * G16V_NENT + (uint32_t)((d < 0 ? -d : d) - 1)) * 20;
        g1_aff q; for (int j = 0; j < 10; j++) { q.x.v[j] = e[j]; q.y.v[j] = e[10 + j]; }
        if (d < 0) q.y = fq_neg(q.y);
        acc = jac_madd(acc, q);
    }
    return acc;
}
// The pairing check e(A, B) e(-L, gamma) e(-C, delta) e(-alpha, beta) == 1 as its three data-dependent Miller loops:
// pair j contributes miller_loop(Q[j], P[j]) when bit j of `present` is set (a pair with a point at infinity contributes 1);
// bit 3: Q[0] holds a finite B whose subgroup membership is still to be checked (g16_b_in_subgroup).
struct G16Pairs { g1_aff P[3]; g2_aff Q[3]; uint32_t present; };
// proof bytes (A || B || C) and the accumulated public-input point L (Jacobian); false: a point fails to parse
// (Lout != nullptr: the Fq2 machine's form -- L stays Jacobian in *Lout, no inversion; P[1] is left unset, bit 1 of `present` says L is finite)
ZKP_HD_NOINLINE inline bool g16_pairs(const G16Vk& vk, const uint8_t proof[256], const g1_jac& L, G16Pairs& o, g1_jac* Lout = nullptr) {
    g1_aff A, C; g2_aff B;
    const int ra = g1_from_ark(A, proof), rb = g2_from_ark(B, proof + 64), rc = g1_from_ark(C, proof + 192);
    o.present = 0;
    if (ra == 0 || rb == 0 || rc == 0) return false;
    if (rb == 1) { o.Q[0] = B; o.present |= 8u; }
    if (ra == 1 && rb == 1) { o.P[0] = A; o.present |= 1u; }
    g1_aff La;
    if (Lout) { *Lout = L; if (!jac_is_inf(L)) { o.Q[1] = vk.gamma; o.present |= 2u; } }
    else if (jac_to_aff(La, L)) { o.P[1] = aff_neg(La); o.Q[1] = vk.gamma; o.present |= 2u; }
    if (rc == 1) { o.P[2] = aff_neg(C); o.Q[2] = vk.delta; o.present |= 4u; }
    return true;
}
// The machine's inputs for the pair (gamma, -L) with L = (X, Y, Z) Jacobian: chain B evaluates a line (a yp, b xp, c) at xp = X / Z^2,
// yp = -Y / Z^3 scaled by Z^3 -- a factor in Fq, which the final exponentiation removes -- as (a (-Y), b (X Z), c Z^3): p = (X Z, -Y), pz = (Z^3, 0)
ZKP_HD inline void g16_vm_pair1(const g1_jac& L, fq2& p, fq2& pz) {
    const fq zz = fq_reduce_weak(fq_sq(L.Z));
    p = fq2{fq_reduce_weak(fq_mul(L.X, L.Z)), fq_neg(L.Y)};
    pz = fq2{fq_reduce_weak(fq_mul(zz, L.Z)), fq_zero()};
}
ZKP_HD inline bool g16_b_in_subgroup(const G16Pairs& o) { return (o.present & 8u) == 0 || g2_in_subgroup(o.Q[0]); }
ZKP_HD inline fq12 g16_pair_miller(const G16Pairs& o, uint32_t j) { return (o.present >> j) & 1u ? miller_loop(o.Q[j], o.P[j]) : fq12_one(); }
ZKP_HD_NOINLINE inline bool g16_finish(const G16Vk& vk, const fq12& f0, const fq12& f1, const fq12& f2) {
    return fq12_is_one(final_exponentiation_chain(fq12_mul(fq12_mul(vk.ml_alpha_beta, f0), fq12_mul(f1, f2))));
}
// ---- envelope headers: what the public inputs are, before any curve arithmetic (shared by the one-lane form below and the 16-lane form)
struct G16Inputs { uint32_t c[8]; uint32_t n; const uint8_t* set; const uint8_t* proof; };      // commitment, set size and bytes (membership), A || B || C
// equality envelope (scheme 2, 298 bytes): public input = the embedded 32-byte commitment as an integer < r
ZKP_HD inline bool g16_equality_header(const G16Vk& vk, const uint8_t* env, uint32_t len, G16Inputs& h) {
    if (len != 298 || env[0] != 2 || env[1] != 2 || vk.n_ic != 2) return false;
    if (ld_u32_le(env + 2) != 256 || ld_u32_le(env + 6) != 32) return false;
    ld_le_words(h.c, env + 266);
    h.n = 0; h.set = nullptr; h.proof = env + 10;
    return fr_raw_lt_r(h.c);
}
// membership envelope (scheme 4): payload = u32 n || n x u64 set || 256-byte proof; public inputs =
// commitment, 64 set slots (zero padded), 64 is_real flags (snark.rs:482-492)
ZKP_HD inline bool g16_membership_header(const G16Vk& vk, const uint8_t* env, uint32_t len, G16Inputs& h) {
    if (len < 10 + 4 + 256 + 32 || env[0] != 2 || env[1] != 4 || vk.n_ic != 2 + 2 * G16_MAX_SET) return false;
    const uint32_t plen = ld_u32_le(env + 2), clen = ld_u32_le(env + 6);
    if (clen != 32 || (uint64_t)10 + plen + clen != len || plen < 4 + 256) return false;
    h.n = ld_u32_le(env + 10);
    if (h.n > G16_MAX_SET || plen != 4 + 8 * h.n + 256) return false;
    ld_le_words(h.c, env + 10 + plen);
    h.set = env + 14; h.proof = env + 14 + 8 * h.n;
    return fr_raw_lt_r(h.c);
}
ZKP_HD inline bool g16_header(int kind, const G16Vk& vk, const uint8_t* env, uint32_t len, G16Inputs& h) {
    return kind == G16_EQUALITY ? g16_equality_header(vk, env, len, h) : g16_membership_header(vk, env, len, h);
}
ZKP_HD inline uint64_t g16_set_element(const G16Inputs& h, uint32_t i) { uint64_t v = 0; for (int k = 0; k < 8; k++) v |= (uint64_t)h.set[8 * i + k] << (8 * k); return v; }
// L = IC_0 + c IC_1 [+ sum_i (v_i IC_{2+i} + IC_{2+64+i})], one lane
ZKP_HD_NOINLINE inline g1_jac g16_public_input_point(const G16Vk& vk, const G16Inputs& h) {
    g1_jac L = jac_add(jac_from_aff(ld_ic(vk, 0)), g16_ic_mul(vk, 1, h.c, G16V_NWIN));
    for (uint32_t i = 0; i < h.n; i++) {
        const uint64_t v = g16_set_element(h, i);
        const uint32_t vw[8] = {(uint32_t)v, (uint32_t)(v >> 32), 0, 0, 0, 0, 0, 0};
        if (v) L = jac_add(L, vk.ic_table ? g16_ic_mul(vk, 2 + i, vw, G16V_NWIN_U64) : g1_mul_u64(ld_ic(vk, 2 + i), v));
        L = jac_madd(L, ld_ic(vk, 2 + G16_MAX_SET + i));            // is_real = 1
    }
    return L;
}
// The same point on `nlanes` cooperating lanes (round 4: VERDICT r03 item 5).  A membership envelope's L is 26 + 8 n + 1 table steps
// (26 windows of the commitment, 7 windows and the is_real point of each set element, IC_0): 155 of them for a 16-element set, one
// after the other on ONE lane while a batch of 1024 envelopes occupied 16 waves of a 1024-wave chip (2.0 ms of the call's 8.5).  Here
// lane t of the envelope takes steps t, t + nlanes, ...; the caller adds the lanes' partial points (any order: the sum is the same
// point).  Digits are the ones g16_ic_mul walks -- a window's digit needs the carries of the windows below it, integer work that every
// lane redoes for its own windows.  Without window tables (host emulation of a large key) lane 0 does it all.
ZKP_HD inline int32_t g16_ic_digit(const uint32_t k[8], uint32_t w) {
    uint32_t carry = 0; int32_t d = 0;
    for (uint32_t j = 0; j <= w; j++) {
        const uint32_t bit = G16V_WBITS * j, wd = bit >> 5, sh = bit & 31u;
        uint32_t x = wd < 8 ? k[wd] >> sh : 0u;
        if (sh + G16V_WBITS > 32 && wd + 1 < 8) x |= k[wd + 1] << (32 - sh);
        const uint32_t dd = (x & ((1u << G16V_WBITS) - 1u)) + carry;
        carry = dd > G16V_NENT ? 1u : 0u;
        d = (int32_t)dd - (int32_t)(carry << G16V_WBITS);
    }
    return d;
}
ZKP_HD inline uint32_t g16_public_input_steps(const G16Inputs& h) { return G16V_NWIN + 1 + (G16V_NWIN_U64 + 1) * h.n; }
ZKP_HD_NOINLINE inline g1_jac g16_public_input_lane(const G16Vk& vk, const G16Inputs& h, uint32_t lane, uint32_t nlanes) {
    if (vk.ic_table == nullptr) return lane == 0 ? g16_public_input_point(vk, h) : jac_infinity<fq>();
    g1_jac acc = jac_infinity<fq>();
    const uint32_t S = g16_public_input_steps(h);
    for (uint32_t s = lane; s < S; s += nlanes) {
        uint32_t ic, w; int32_t d;
        if (s < G16V_NWIN) { ic = 1; w = s; d = g16_ic_digit(h.c, w); }
        else if (s == G16V_NWIN) { acc = jac_madd(acc, ld_ic(vk, 0)); continue; }
        else {
            const uint32_t q = s - G16V_NWIN - 1, i = q / (G16V_NWIN_U64 + 1); w = q % (G16V_NWIN_U64 + 1);
            if (w == G16V_NWIN_U64) { acc = jac_madd(acc, ld_ic(vk, 2 + G16_MAX_SET + i)); continue; }      // is_real = 1
            const uint64_t v = g16_set_element(h, i);
            const uint32_t vw[8] = {(uint32_t)v, (uint32_t)(v >> 32), 0, 0, 0, 0, 0, 0};
            ic = 2 + i; d = g16_ic_digit(vw, w);
        }
        if (d == 0) continue;
        const uint32_t* e = vk.ic_table + (((size_t)ic * G16V_NWIN + w) * G16V_NENT + (uint32_t)((d < 0 ? -d : d) - 1)) * 20;
        g1_aff q; for (int j = 0; j < 10; j++) { q.x.v[j] = e[j]; q.y.v[j] = e[10 + j]; }
        if (d < 0) q.y = fq_neg(q.y);
        acc = jac_madd(acc, q);
    }
    return acc;
}
// Lin != nullptr: the public-input point was accumulated elsewhere (g16_public_input_lane)
ZKP_HD_NOINLINE inline bool g16_equality_pairs(const G16Vk& vk, const uint8_t* env, uint32_t len, G16Pairs& o, g1_jac* Lout = nullptr, const g1_jac* Lin = nullptr) {
    o.present = 0;
    G16Inputs h;
    if (!g16_equality_header(vk, env, len, h)) return false;
    return g16_pairs(vk, h.proof, Lin ? *Lin : g16_public_input_point(vk, h), o, Lout);
}
ZKP_HD_NOINLINE inline bool g16_membership_pairs(const G16Vk& vk, const uint8_t* env, uint32_t len, G16Pairs& o, g1_jac* Lout = nullptr, const g1_jac* Lin = nullptr) {
    o.present = 0;
    G16Inputs h;
    if (!g16_membership_header(vk, env, len, h)) return false;
    return g16_pairs(vk, h.proof, Lin ? *Lin : g16_public_input_point(vk, h), o, Lout);
}
// one envelope start to finish in one thread (what the three GPU kernels compute between them; used by the host emulation)
ZKP_HD inline bool g16_verify_envelope(int kind, const G16Vk& vk, const uint8_t* env, uint32_t len) {
    G16Pairs o;
    if (!(kind == G16_EQUALITY ? g16_equality_pairs(vk, env, len, o) : g16_membership_pairs(vk, env, len, o))) return false;
    if (!g16_b_in_subgroup(o)) return false;
    return g16_finish(vk, g16_pair_miller(o, 0), g16_pair_miller(o, 1), g16_pair_miller(o, 2));
}

}  // namespace zkp
