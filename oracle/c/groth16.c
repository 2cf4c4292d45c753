/* ORACLE (test infrastructure only; never linked into or called by the product path).
 *
 * CPU restatement of libzkp's SNARK proving path in plain C: MiMC commitments, the equality and set-membership circuits,
 * the Groth16 prover and libzkp's envelope framing.  The reference reaches the prover through the third-party crates
 * ark-groth16 / ark-r1cs-std / ark-relations / ark-serialize ^0.5 (Cargo.toml:16-27; not vendored, not pinned, not
 * buildable here: no cargo/rustc).  PARITY UNPINNED at the proof-byte level (OsRng setup, OsRng r and s, no byte vectors in
 * the reference's tests); this file is pinned to oracle/py/groth16.py bit for bit on committed vectors
 * (tests/test_oracle_c_snark.py), and that model is pinned by the pairing check.
 *
 * The algorithm is upstream's (what bench.py's cpu_baseline times): constraint synthesis row by row, the QAP witness map
 * with radix-2 FFTs on the domain and its coset (LibsnarkReduction), and five variable-base bucket MSMs -- not the GPU
 * library's fixed-base window tables.
 *
 * Reference lines followed:
 *   snark.rs:186-221  MiMC constants, mimc_hash_native, fr_to_commitment       snark.rs:232-247  mimc_hash_circuit
 *   snark.rs:262-291  EqualityCircuit::generate_constraints                    snark.rs:514-585  MembershipCircuit
 *   snark.rs:343-374  prove_equality_zk (256-byte uncompressed proof)          snark.rs:405-452  prove_membership_zk
 *   equality_proof.rs:10-32, set_membership.rs:12-38, proof/mod.rs:23-36       validation and envelopes
 */
#include "zkp_oracle.h"
#include "bn254.h"
#include "transcript.h"
#include <string.h>
#include <stdlib.h>

#define MIMC_ROUNDS 110
#define MAX_SET 64

static fp MIMC_C[MIMC_ROUNDS];
static int g_snark_init = 0;

typedef struct {
    int loaded;
    uint32_t n_inst, n_wit, nv, m;
    g1a alpha_g1, beta_g1, delta_g1; g2a beta_g2, delta_g2;
    g1a *a_query, *b_g1_query, *h_query, *l_query; g2a* b_g2_query;
} g16_key;
static g16_key KEY[2];

static void snark_init(void) {
    if (g_snark_init) return;
    bn254_init();
    for (uint32_t i = 0; i < MIMC_ROUNDS; i++) {                 /* snark.rs:186-199 */
        uint8_t in[23], h[32];
        memcpy(in, "libzkp_mimc_v1:", 15);
        for (int k = 0; k < 8; k++) in[15 + k] = (uint8_t)((uint64_t)i >> (8 * k));
        sha256(h, in, 23);
        fr_from_bytes_mod_order(&MIMC_C[i], h);
    }
    g_snark_init = 1;
}

/* snark.rs:201-211 */
static void mimc_native(fp* out, uint64_t value) {
    fp x, t, t2; fr_from_u64(&x, value);
    for (int i = 0; i < MIMC_ROUNDS; i++) {
        fr_add(&t, &x, &MIMC_C[i]);
        fr_mul(&t2, &t, &t); fr_mul(&t2, &t2, &t2); fr_mul(&x, &t2, &t);
    }
    *out = x;
}
int zkp_oracle_snark_commit_value(uint64_t value, uint8_t out[32]) {     /* utils/commitment.rs:14-16 */
    snark_init();
    fp h; mimc_native(&h, value); fr_to_bytes(out, &h);
    return 0;
}

/* ---------------------------------------------------------------- key file (ark-serialize uncompressed ProvingKey<Bn254>) */
typedef struct { const uint8_t* p; uint64_t left; int ok; } reader;
static const uint8_t* rd(reader* r, uint64_t n) { if (r->left < n) { r->ok = 0; return NULL; } const uint8_t* q = r->p; r->p += n; r->left -= n; return q; }
static uint64_t rd_u64(reader* r) { const uint8_t* q = rd(r, 8); uint64_t v = 0; if (q) for (int i = 0; i < 8; i++) v |= (uint64_t)q[i] << (8 * i); return v; }
static int rd_g1(reader* r, g1a* out) { const uint8_t* q = rd(r, 64); return q && g1_parse(out, q); }
static int rd_g2(reader* r, g2a* out) { const uint8_t* q = rd(r, 128); return q && g2_parse(out, q); }
static g1a* rd_vec_g1(reader* r, uint64_t* n) {
    *n = rd_u64(r); if (!r->ok || *n > (1u << 24)) { r->ok = 0; return NULL; }
    g1a* v = (g1a*)malloc(sizeof(g1a) * (*n + 1));
    for (uint64_t i = 0; i < *n; i++) if (!rd_g1(r, &v[i])) { r->ok = 0; free(v); return NULL; }
    return v;
}
static g2a* rd_vec_g2(reader* r, uint64_t* n) {
    *n = rd_u64(r); if (!r->ok || *n > (1u << 24)) { r->ok = 0; return NULL; }
    g2a* v = (g2a*)malloc(sizeof(g2a) * (*n + 1));
    for (uint64_t i = 0; i < *n; i++) if (!rd_g2(r, &v[i])) { r->ok = 0; free(v); return NULL; }
    return v;
}
/* ProvingKey { vk { alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1 }, beta_g1, delta_g1, a_query, b_g1_query,
 * b_g2_query, h_query, l_query }; kind 0 = equality_mimc, 1 = membership_mimc (snark.rs:306,327) */
int zkp_oracle_g16_load_key(int kind, const uint8_t* pk, uint64_t len) {
    snark_init();
    if (kind < 0 || kind > 1 || !pk) return ZKP_ORACLE_INVALID_INPUT;
    g16_key* K = &KEY[kind];
    if (K->loaded) { free(K->a_query); free(K->b_g1_query); free(K->h_query); free(K->l_query); free(K->b_g2_query); memset(K, 0, sizeof *K); }
    reader R = {pk, len, 1};
    g2a gamma_g2; uint64_t n_abc = 0, na = 0, nb1 = 0, nb2 = 0, nh = 0, nl = 0;
    int ok = rd_g1(&R, &K->alpha_g1) && rd_g2(&R, &K->beta_g2) && rd_g2(&R, &gamma_g2) && rd_g2(&R, &K->delta_g2);
    g1a* abc = ok ? rd_vec_g1(&R, &n_abc) : NULL;
    ok = ok && abc && rd_g1(&R, &K->beta_g1) && rd_g1(&R, &K->delta_g1);
    if (ok) K->a_query = rd_vec_g1(&R, &na);
    if (ok && K->a_query) K->b_g1_query = rd_vec_g1(&R, &nb1);
    if (ok && K->b_g1_query) K->b_g2_query = rd_vec_g2(&R, &nb2);
    if (ok && K->b_g2_query) K->h_query = rd_vec_g1(&R, &nh);
    if (ok && K->h_query) K->l_query = rd_vec_g1(&R, &nl);
    free(abc);
    const uint32_t want_inst = kind == 0 ? 2 : 130, want_wit = kind == 0 ? 332 : 523, want_m = kind == 0 ? 512 : 1024;
    if (!ok || !R.ok || R.left != 0 || !K->l_query || n_abc != want_inst || na != want_inst + want_wit || nb1 != na || nb2 != na || nh != want_m - 1 || nl != want_wit) {
        free(K->a_query); free(K->b_g1_query); free(K->h_query); free(K->l_query); free(K->b_g2_query); memset(K, 0, sizeof *K);
        return ZKP_ORACLE_INVALID_PROOF_FORMAT;
    }
    K->n_inst = want_inst; K->n_wit = want_wit; K->nv = (uint32_t)na; K->m = want_m;
    K->loaded = 1;
    return 0;
}

/* ---------------------------------------------------------------- constraint synthesis
 * The prover needs the full assignment z = (instance | witness) and the row evaluations a_j = <A_j, z>, b_j, c_j; each
 * helper allocates / enforces exactly what the corresponding ark-r1cs-std call does, in the reference's order. */
typedef struct {
    fp inst[130], wit[523];
    uint32_t n_inst, n_wit, n_rows;
    fp a[1024], b[1024], c[1024];
} synth;
static void cs_row(synth* S, const fp* a, const fp* b, const fp* c) { S->a[S->n_rows] = *a; S->b[S->n_rows] = *b; S->c[S->n_rows] = *c; S->n_rows++; }
static fp cs_mul(synth* S, const fp* a, const fp* b) {         /* AllocatedFp::mul: new witness + a * b = product */
    fp p; fr_mul(&p, a, b);
    S->wit[S->n_wit++] = p;
    cs_row(S, a, b, &p);
    return p;
}
static void cs_enforce_equal(synth* S, const fp* a, const fp* b) { fp d; fr_sub(&d, a, b); cs_row(S, &d, &FR_ONE, &FR_ZERO); }    /* (a - b) * 1 = 0 */
static fp cs_mimc(synth* S, fp x) {                                 /* snark.rs:232-247 */
    for (int i = 0; i < MIMC_ROUNDS; i++) {
        fp t; fr_add(&t, &x, &MIMC_C[i]);
        const fp t2 = cs_mul(S, &t, &t), t4 = cs_mul(S, &t2, &t2);
        x = cs_mul(S, &t4, &t);
    }
    return x;
}
static void synth_equality(synth* S, uint64_t a, uint64_t b, const fp* commitment) {      /* snark.rs:262-291 */
    memset(S, 0, sizeof *S);
    S->inst[0] = FR_ONE; S->n_inst = 1;
    fp av, bv; fr_from_u64(&av, a); fr_from_u64(&bv, b);
    S->wit[S->n_wit++] = av; S->wit[S->n_wit++] = bv;
    cs_enforce_equal(S, &av, &bv);
    const fp h = cs_mimc(S, av);
    S->inst[S->n_inst++] = *commitment;
    cs_enforce_equal(S, &h, commitment);
}
static void synth_membership(synth* S, uint64_t value, const uint64_t* set, uint32_t count, uint32_t pos, const fp* commitment) {   /* snark.rs:514-585 */
    memset(S, 0, sizeof *S);
    S->inst[0] = FR_ONE; S->n_inst = 1;
    fp v; fr_from_u64(&v, value);
    S->wit[S->n_wit++] = v;
    const fp h = cs_mimc(S, v);
    S->inst[S->n_inst++] = *commitment;
    cs_enforce_equal(S, &h, commitment);
    fp setv[MAX_SET], real[MAX_SET], sel[MAX_SET];
    for (uint32_t i = 0; i < MAX_SET; i++) { fr_from_u64(&setv[i], i < count ? set[i] : 0); S->inst[S->n_inst++] = setv[i]; }
    for (uint32_t i = 0; i < MAX_SET; i++) {                         /* Boolean::new_input: (1 - b) * b = 0 */
        real[i] = i < count ? FR_ONE : FR_ZERO; S->inst[S->n_inst++] = real[i];
        fp nb; fr_sub(&nb, &FR_ONE, &real[i]); cs_row(S, &nb, &real[i], &FR_ZERO);
    }
    for (uint32_t i = 0; i < MAX_SET; i++) {                         /* Boolean::new_witness likewise */
        sel[i] = i == pos ? FR_ONE : FR_ZERO; S->wit[S->n_wit++] = sel[i];
        fp nb; fr_sub(&nb, &FR_ONE, &sel[i]); cs_row(S, &nb, &sel[i], &FR_ZERO);
    }
    fp total = FR_ZERO;
    for (uint32_t i = 0; i < MAX_SET; i++) {
        fr_add(&total, &total, &sel[i]);
        fp nr; fr_sub(&nr, &FR_ONE, &real[i]);
        const fp prod = cs_mul(S, &sel[i], &nr);
        cs_enforce_equal(S, &prod, &FR_ZERO);
    }
    cs_enforce_equal(S, &total, &FR_ONE);
    fp acc = FR_ZERO;
    for (uint32_t i = 0; i < MAX_SET; i++) {
        fp d; fr_sub(&d, &v, &setv[i]);
        const fp p = cs_mul(S, &sel[i], &d);
        fr_add(&acc, &acc, &p);
    }
    cs_enforce_equal(S, &acc, &FR_ZERO);
}

/* ---------------------------------------------------------------- radix-2 FFT over Fr (in place, natural order in and out) */
static void fft(fp* a, uint32_t n, const fp* w) {
    for (uint32_t i = 1, j = 0; i < n; i++) {
        uint32_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { const fp t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (uint32_t len = 2; len <= n; len <<= 1) {
        fp wl; fr_pow_u64(&wl, w, n / len);
        for (uint32_t s = 0; s < n; s += len) {
            fp x = FR_ONE;
            for (uint32_t k = 0; k < len / 2; k++) {
                fp u = a[s + k], v; fr_mul(&v, &a[s + k + len / 2], &x);
                fr_add(&a[s + k], &u, &v); fr_sub(&a[s + k + len / 2], &u, &v);
                fr_mul(&x, &x, &wl);
            }
        }
    }
}
static void ifft(fp* a, uint32_t n, const fp* w) {
    fp wi, ni, nn; fr_inv(&wi, w); fr_from_u64(&nn, n); fr_inv(&ni, &nn);
    fft(a, n, &wi);
    for (uint32_t i = 0; i < n; i++) fr_mul(&a[i], &a[i], &ni);
}
static void scale_powers(fp* a, uint32_t n, const fp* g) { fp x = FR_ONE; for (uint32_t i = 0; i < n; i++) { fr_mul(&a[i], &a[i], &x); fr_mul(&x, &x, g); } }
/* LibsnarkReduction::witness_map_from_matrices: h = (a * b - c) / Z on the coset g * <w>, g = 5 */
static void witness_map(fp* h, synth* S, uint32_t m) {
    fp w; fr_root_of_unity(&w, m);
    fp *a = S->a, *b = S->b, *c = S->c;
    for (uint32_t i = S->n_rows; i < m; i++) { a[i] = FR_ZERO; b[i] = FR_ZERO; c[i] = FR_ZERO; }
    for (uint32_t i = 0; i < S->n_inst; i++) a[S->n_rows + i] = S->inst[i];
    fp g, gi; fr_from_u64(&g, 5); fr_inv(&gi, &g);
    fp* v[3] = {a, b, c};
    for (int k = 0; k < 3; k++) { ifft(v[k], m, &w); scale_powers(v[k], m, &g); fft(v[k], m, &w); }
    fp zinv; fr_pow_u64(&zinv, &g, m); fr_sub(&zinv, &zinv, &FR_ONE); fr_inv(&zinv, &zinv);
    for (uint32_t i = 0; i < m; i++) { fp t; fr_mul(&t, &a[i], &b[i]); fr_sub(&t, &t, &c[i]); fr_mul(&h[i], &t, &zinv); }
    ifft(h, m, &w); scale_powers(h, m, &gi);
}

/* project tape (oracle/py/groth16.py: draw_fr) */
static void draw_fr(fp* r, const uint8_t seed[32], uint32_t idx, uint32_t slot) {
    uint8_t b[64]; zkp_oracle_tape_draw64(seed, idx, slot, b); fr_from_bytes_wide(r, b);
}

/* create_proof_with_reduction: A = alpha + sum z_i a_i + r delta; B likewise in G2 (and G1); C = sum aux l + sum h H + s A + r B1 - rs delta */
static int g16_prove(const g16_key* K, synth* S, const uint8_t seed[32], uint8_t proof[256]) {
    if (!K->loaded || S->n_inst != K->n_inst || S->n_wit != K->n_wit) return ZKP_ORACLE_PROOF_GENERATION_FAILED;
    fp r, s; draw_fr(&r, seed, 0x47313600u, 0); draw_fr(&s, seed, 0x47313600u, 1);
    fp* h = (fp*)malloc(sizeof(fp) * K->m);
    witness_map(h, S, K->m);
    fp* z = (fp*)malloc(sizeof(fp) * K->nv);
    memcpy(z, S->inst, sizeof(fp) * K->n_inst); memcpy(z + K->n_inst, S->wit, sizeof(fp) * K->n_wit);
    g1j A, B1, C, t; g2j B2, t2;
    /* z[0] = 1: query[0] is added as is, the rest goes through the MSM (ark's calculate_coeff) */
    g1_msm(&A, K->nv - 1, z + 1, K->a_query + 1); g1j_madd(&A, &A, &K->a_query[0]);
    g1j_from_affine(&t, &K->delta_g1); g1j_mul(&t, &t, &r); g1j_add(&A, &A, &t); g1j_madd(&A, &A, &K->alpha_g1);
    g1_msm(&B1, K->nv - 1, z + 1, K->b_g1_query + 1); g1j_madd(&B1, &B1, &K->b_g1_query[0]);
    g1j_from_affine(&t, &K->delta_g1); g1j_mul(&t, &t, &s); g1j_add(&B1, &B1, &t); g1j_madd(&B1, &B1, &K->beta_g1);
    g2_msm(&B2, K->nv - 1, z + 1, K->b_g2_query + 1); g2j_madd(&B2, &B2, &K->b_g2_query[0]);
    g2j_from_affine(&t2, &K->delta_g2); g2j_mul(&t2, &t2, &s); g2j_add(&B2, &B2, &t2); g2j_madd(&B2, &B2, &K->beta_g2);
    g1j hacc, lacc;
    g1_msm(&hacc, K->m - 1, h, K->h_query);
    g1_msm(&lacc, K->n_wit, S->wit, K->l_query);
    g1j sA, rB; g1j_mul(&sA, &A, &s); g1j_mul(&rB, &B1, &r);
    fp rs; fr_mul(&rs, &r, &s);
    g1j_from_affine(&t, &K->delta_g1); g1j_mul(&t, &t, &rs); g1j_neg(&t, &t);
    g1j_add(&C, &sA, &rB); g1j_add(&C, &C, &t); g1j_add(&C, &C, &lacc); g1j_add(&C, &C, &hacc);
    g1a Aa, Ca; g2a Ba;
    g1j_to_affine(&Aa, &A); g2j_to_affine(&Ba, &B2); g1j_to_affine(&Ca, &C);
    g1_serialize(proof, &Aa); g2_serialize(proof + 64, &Ba); g1_serialize(proof + 192, &Ca);     /* snark.rs:369-373 */
    free(h); free(z);
    return 0;
}

static void put_u32(uint8_t* p, uint32_t x) { for (int i = 0; i < 4; i++) p[i] = (uint8_t)(x >> (8 * i)); }

/* proof::equality_proof::prove_equality (equality_proof.rs:10-32) -> 298-byte envelope (scheme 2) */
int zkp_oracle_prove_equality(uint64_t val1, uint64_t val2, const uint8_t seed[32], uint8_t* out, uint32_t cap, uint32_t* out_len) {
    snark_init();
    *out_len = 0;
    if (val1 != val2) return ZKP_ORACLE_INVALID_INPUT;                    /* validation.rs:21-27 */
    if (cap < 298) return ZKP_ORACLE_BUFFER_TOO_SMALL;
    fp c; mimc_native(&c, val1);
    synth* S = (synth*)malloc(sizeof(synth));
    synth_equality(S, val1, val2, &c);
    out[0] = 2; out[1] = 2; put_u32(out + 2, 256); put_u32(out + 6, 32);
    const int rc = g16_prove(&KEY[0], S, seed, out + 10);
    free(S);
    if (rc) return rc;
    fr_to_bytes(out + 266, &c);
    *out_len = 298;
    return 0;
}

/* proof::set_membership::prove_membership (set_membership.rs:12-38): payload = u32 |set| || set || groth16 proof (scheme 4) */
int zkp_oracle_prove_membership(uint64_t value, const uint64_t* set, uint32_t count, const uint8_t seed[32], uint8_t* out, uint32_t cap, uint32_t* out_len) {
    snark_init();
    *out_len = 0;
    if (count == 0 || count > MAX_SET) return ZKP_ORACLE_INVALID_INPUT;   /* validation.rs:50-63, snark.rs:406-418 */
    uint32_t pos = count;
    for (uint32_t i = 0; i < count; i++) if (set[i] == value) { pos = i; break; }
    if (pos == count) return ZKP_ORACLE_INVALID_INPUT;
    const uint32_t plen = 4 + 8 * count + 256, total = 10 + plen + 32;
    if (cap < total) return ZKP_ORACLE_BUFFER_TOO_SMALL;
    fp c; mimc_native(&c, value);
    synth* S = (synth*)malloc(sizeof(synth));
    synth_membership(S, value, set, count, pos, &c);
    out[0] = 2; out[1] = 4; put_u32(out + 2, plen); put_u32(out + 6, 32);
    put_u32(out + 10, count);
    for (uint32_t i = 0; i < count; i++) for (int k = 0; k < 8; k++) out[14 + 8 * i + k] = (uint8_t)(set[i] >> (8 * k));
    const int rc = g16_prove(&KEY[1], S, seed, out + 14 + 8 * count);
    free(S);
    if (rc) return rc;
    fr_to_bytes(out + 10 + plen, &c);
    *out_len = total;
    return 0;
}
