// Per-thread steps of the batched Bulletproofs single-value range prover (n = 8, 16, 32 or 64 bits per batch; m = 1).
//
// What is computed follows bulletproofs::RangeProof::prove_single as the reference calls it
// (/root/reference/src/backend/bulletproofs.rs:138-158,344-352,396-404,643-653; protocol restated in
// SURVEY.md appendix A.3 and oracle/py/bulletproofs.py).  HOW it is computed is MI355X-first:
//   * every multiscalar multiplication is over the 130 FIXED generators (B, B~, G_0..63, H_0..63): the
//     inner-product rounds never fold generator points; instead per-generator coefficients g_i, h_i are
//     folded (scalar work) and each L_k / R_k is a 65-term fixed-base MSM with no doublings;
//   * work is laid out structure-of-arrays with the proof index ("job") fastest so that one lane = one
//     proof and every global access is coalesced across the wavefront;
//   * each function below is the body of one kernel thread (bp_kernels.hip wraps them); they are
//     host+device so tests/emul can run the identical code on the CPU against the oracle.
#pragma once
#include "fe25519.h"
#include "sc25519.h"
#include "ge25519.h"
#include "keccak.h"

namespace zkp {

constexpr uint32_t BP_N = 64;          // widest proof; workspace strides and generator tables are sized for it
constexpr uint32_t BASE_B = 0, BASE_BB = 1, BASE_G = 2, BASE_H = 66, NBASE = 130;
// fixed-base tables of the 130 generators: signed radix-1024 digits -> 26 windows of 512 affine-niels entries (208 MB);
// a 64-bit value needs 7 windows.  (The Groth16 key tables use radix 256: G16_* in g16_steps.h.)
constexpr uint32_t WBITS = 10, NWIN = 26, NENT = 512, DIGW = 13, NWIN_U64 = 7, NIELS_W = 30, SUBTAB_W = NENT * NIELS_W;
constexpr uint32_t TAPE_SLOTS = 132;
// phase-1 MSM slots: V = v*B + gamma*B~ ; A = a_bl*B~ + sum bit_i*G_i + (bit_i-1)*H_i ; S = s_bl*B~ + sum sL_i*G_i + sR_i*H_i
// Slot numbers are compact for the batch's bit width n (so a launch's slot index is also its digit row): A's slots start at
// P1_A, S's at p1_s(n) = 3 + 2n; the capacities below are those of n = 64.
constexpr uint32_t P1_V = 0, P1_A = 2, P1_NSLOTS = 260;
constexpr uint32_t P2_NSLOTS = 4;      // T1: t1*B + t1_bl*B~ ; T2: t2*B + t2_bl*B~
constexpr uint32_t PR_NSLOTS = 130;    // L: c_L*w*B + n/2 G + n/2 H ; R likewise  (slot n+1 starts R)
ZKP_HD constexpr uint32_t p1_s(uint32_t n) { return 3 + 2 * n; }
ZKP_HD constexpr uint32_t tape_slots(uint32_t n) { return 2 * n + 4; }
ZKP_HD constexpr uint32_t rp_bytes(uint32_t lg) { return 32 * (9 + 2 * lg); }     // RangeProof::to_bytes: 9 + 2 lg n elements
constexpr uint32_t GE_W = 40;
enum { SC_Y = 0, SC_Z, SC_X, SC_W, SC_U, SC_UINV, SC_T0, SC_T1, SC_T2, SC_NUM };
enum { KIND_RANGE_MIN = 0, KIND_RANGE_MAX, KIND_THRESHOLD, KIND_CONSISTENCY, KIND_BULLETPROOF, KIND_NUM };

struct BpView {
    uint32_t M;                    // number of proof jobs
    uint32_t n, lg;                // bits per proof (8, 16, 32, 64) and log2 of it: one width per batch
    uint32_t dig16 = 0;            // 1: MSM digits in signed radix 2^16 (the HBM-resident tables of edg.h: the device prover); 0: radix 1024
    // job description (read-only)
    const uint64_t* v;             // [M] value in [0, 2^64)
    const uint32_t* seed_ix;       // [M] which 32-byte seed
    const uint32_t* proof_ix;      // [M] tape proof index
    const int32_t* bl_plus;        // [M] gamma = +blinding[bl_plus] (if >= 0) - blinding[bl_minus] (if >= 0)
    const int32_t* bl_minus;       // [M]
    const uint8_t* kind;           // [M] transcript label
    const uint32_t* seeds;         // [nseeds][8]
    const uint64_t* proof_off;     // [M] byte offset of the 672-byte RangeProof in out
    const uint64_t* commit_off;    // [M] byte offset of the 32-byte commitment V in out
    uint8_t* out;
    // workspace, all [..][8][M] words unless noted
    uint32_t* tape;                // [132]
    uint32_t* gamma;               // [1]
    uint32_t* d1;                  // [260] packed signed digits
    uint32_t* d2;                  // [4]
    uint32_t* dr;                  // [130]
    uint32_t* yinvpow;             // [64] y^-i (written by step_poly)
    uint32_t* ypq;                 // [32]: y^b (b < 8), y^(8a) (a < 8), then the same for y^-1: y^i = y^(8a) * y^b
    uint32_t* r0;                  // [64]
    uint32_t* r1;                  // [64]
    uint32_t* pp;                  // [3*64] products
    uint32_t* ab;                  // [2 buffers][2][64]
    uint32_t* gh;                  // [2][64]
    uint32_t* scal;                // [SC_NUM]
    uint32_t* tstate;              // [52][M] words: STROBE state, pos, pos_begin
    uint32_t* enc;                 // [3][8][M] encodings of the current phase's points
};

ZKP_HD inline sc ld_sc(const uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows) {
    sc r; const uint32_t* q = p + (size_t)idx * 8 * rows + row;
    ZKP_UNROLL for (int k = 0; k < 8; k++) r.v[k] = q[(size_t)k * rows];
    return r;
}
ZKP_HD inline void st_sc(uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows, const sc& s) {
    uint32_t* q = p + (size_t)idx * 8 * rows + row;
    ZKP_UNROLL for (int k = 0; k < 8; k++) q[(size_t)k * rows] = s.v[k];
}
// store the signed digits of a raw / Montgomery-form scalar: a row of DIGW words per (slot, row), word-major.  dig16 = 0: radix 1024,
// 26 digits in 13 words; dig16 = 1: radix 2^16 (sc_recode_signed65536: a word of the scalar is two digits), 16 digits in the first 8 words
ZKP_HD inline void st_digits_raw(uint32_t* d, uint32_t slot, uint32_t row, uint32_t rows, const sc& raw, uint32_t dig16 = 0) {
    uint32_t* q = d + (size_t)slot * DIGW * rows + row;
    if (dig16) {
        uint32_t pk[8]; sc_recode_signed65536(pk, raw);
        ZKP_UNROLL for (uint32_t k = 0; k < 8; k++) q[(size_t)k * rows] = pk[k];
        return;
    }
    uint32_t pk[DIGW]; sc_recode_signed1024(pk, raw);
    ZKP_UNROLL for (uint32_t k = 0; k < DIGW; k++) q[(size_t)k * rows] = pk[k];
}
ZKP_HD inline void st_digits(uint32_t* d, uint32_t slot, uint32_t row, uint32_t rows, const sc& mont, uint32_t dig16 = 0) { st_digits_raw(d, slot, row, rows, sc_to_raw(mont), dig16); }
ZKP_HD inline int32_t ld_digit(const uint32_t* d, uint32_t srow, uint32_t w, uint32_t row, uint32_t rows) {
    return (int32_t)(int16_t)(d[((size_t)srow * DIGW + (w >> 1)) * rows + row] >> (16 * (w & 1u)));
}
ZKP_HD inline void put_bytes(uint8_t* dst, const uint32_t* w, int nwords) {
    for (int i = 0; i < nwords; i++) { dst[4 * i] = (uint8_t)w[i]; dst[4 * i + 1] = (uint8_t)(w[i] >> 8); dst[4 * i + 2] = (uint8_t)(w[i] >> 16); dst[4 * i + 3] = (uint8_t)(w[i] >> 24); }
}
ZKP_HD inline void ld_seed(uint32_t s[8], const BpView& V, uint32_t job) {
    const uint32_t* p = V.seeds + (size_t)V.seed_ix[job] * 8;
    ZKP_UNROLL for (int k = 0; k < 8; k++) s[k] = p[k];
}
// libzkp-level blinding i of a seed: from_bytes_mod_order(draw64(seed, 0xFFFFFFFF, i)[0:32])  (bulletproofs.rs:82-87)
ZKP_HD inline sc tape_blinding(const uint32_t seed[8], uint32_t i) {
    uint32_t w[16]; tape_draw64(w, seed, 0xFFFFFFFFu, i);
    sc raw; ZKP_UNROLL for (int k = 0; k < 8; k++) raw.v[k] = w[k];
    return sc_from_raw256(raw);
}

// ------------------------------------------------------------------------------------------------
// step 0: randomness tape + phase-1 digits.  thread = (slot in [0, 2n + 5), job)
ZKP_HD inline void step_tape(const BpView& V, uint32_t slot, uint32_t job) {
    const uint32_t M = V.M, n = V.n, P1_S = p1_s(V.n);
    uint32_t seed[8]; ld_seed(seed, V, job);
    if (slot < tape_slots(n)) {
        uint32_t w[16]; tape_draw64(w, seed, V.proof_ix[job], slot);
        const sc x = sc_from_wide(w);
        st_sc(V.tape, slot, job, M, x);
        if (slot == 0) st_digits(V.d1, P1_A + 0, job, M, x, V.dig16);                            // a_blinding * B~
        else if (slot == 1) st_digits(V.d1, P1_S + 0, job, M, x, V.dig16);                       // s_blinding * B~
        else if (slot < 2 + n) st_digits(V.d1, P1_S + 1 + (slot - 2), job, M, x, V.dig16);    // s_L[i] * G_i
        else if (slot < 2 + 2 * n) st_digits(V.d1, P1_S + 1 + n + (slot - 2 - n), job, M, x, V.dig16);  // s_R[i] * H_i
        else if (slot == 2 + 2 * n) st_digits(V.d2, 1, job, M, x, V.dig16);                   // t1_blinding * B~
        else st_digits(V.d2, 3, job, M, x, V.dig16);                                             // t2_blinding * B~
    } else {
        sc g = sc_zero();
        if (V.bl_plus[job] >= 0) g = tape_blinding(seed, (uint32_t)V.bl_plus[job]);
        if (V.bl_minus[job] >= 0) g = sc_sub(g, tape_blinding(seed, (uint32_t)V.bl_minus[job]));
        st_sc(V.gamma, 0, job, M, g);
        st_digits(V.d1, P1_V + 1, job, M, g, V.dig16);
        const uint64_t v = V.v[job];
        st_digits_raw(V.d1, P1_V + 0, job, M, sc_words((uint32_t)v, (uint32_t)(v >> 32), 0, 0, 0, 0, 0, 0), V.dig16);
        // A: bit_i * G_i + (bit_i - 1) * H_i  -> single-window digits (+1 / 0 and 0 / -1)
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t bit = (uint32_t)(v >> i) & 1u;
            V.d1[(size_t)(P1_A + 1 + i) * DIGW * M + job] = bit;
            V.d1[(size_t)(P1_A + 1 + n + i) * DIGW * M + job] = bit ? 0u : 0xFFFFu;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fixed-base MSM, reference per-thread form (table read straight from global memory).
// table: [NBASE][NWIN][NENT][30] words, entry e = (e+1) * 1024^w * Base in affine niels form.
ZKP_HD inline ge_niels ld_niels(const uint32_t* p) {
    ge_niels n;
    ZKP_UNROLL for (int k = 0; k < 10; k++) { n.ypx.v[k] = p[k]; n.ymx.v[k] = p[10 + k]; n.xy2d.v[k] = p[20 + k]; }
    return n;
}
ZKP_HD inline ge msm_accumulate_digit(const ge& acc, int32_t d, const uint32_t* subtab) {
    const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
    ge_niels n = ld_niels(subtab + (size_t)(mag - 1) * NIELS_W);
    n = ge_niels_select(d < 0, ge_niels_neg(n), n);
    return ge_madd(acc, n);
}
ZKP_HD inline void st_ge(uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows, const ge& g) {
    uint32_t* q = p + (size_t)idx * GE_W * rows + row;
    ZKP_UNROLL for (int k = 0; k < 10; k++) {
        q[(size_t)k * rows] = g.X.v[k]; q[(size_t)(10 + k) * rows] = g.Y.v[k];
        q[(size_t)(20 + k) * rows] = g.Z.v[k]; q[(size_t)(30 + k) * rows] = g.T.v[k];
    }
}
ZKP_HD inline ge ld_ge(const uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows) {
    ge g; const uint32_t* q = p + (size_t)idx * GE_W * rows + row;
    ZKP_UNROLL for (int k = 0; k < 10; k++) {
        g.X.v[k] = q[(size_t)k * rows]; g.Y.v[k] = q[(size_t)(10 + k) * rows];
        g.Z.v[k] = q[(size_t)(20 + k) * rows]; g.T.v[k] = q[(size_t)(30 + k) * rows];
    }
    return g;
}
struct MsmView {
    uint32_t rows, nslots, nchunks;
    const uint32_t* table;       // generator tables
    const uint32_t* digits;      // [nslots][digit words][rows] (radix 1024: 13 words, radix 256: 8 words per scalar)
    const uint16_t* slot_base;   // [nslots] which table (generator / key point) a slot uses
    const uint16_t* slot_scalar; // [nslots] which digit row a slot reads (nullptr: row = slot)
    const uint8_t* slot_nwin;    // [nslots] number of low windows that may be non-zero
    const uint16_t* chunk_begin; // [nchunks] first slot of a chunk
    const uint16_t* chunk_win0;  // [nchunks] first window inside that slot (chunks are window-granular)
    const uint16_t* chunk_nwin;  // [nchunks] number of (slot, window) steps in the chunk
    uint32_t* partial;           // [nchunks][40][rows]
    const uint32_t* acc_init;    // optional [ACC_W]: every chunk's accumulator starts from this point (Weierstrass MSMs: a
                                 // fixed offset point so the unchecked mixed addition never sees infinity); nullptr = identity
    uint32_t nwin = 0, nent = 0, digw = 0;      // k_msm_gather only: shape of the table behind `table` (windows per point, entries per
                                                // window, digit words per scalar) -- a run-time property of the loaded Groth16 key
    uint32_t slot_ent = 0, uneven = 0;          // entries of one point's block; uneven radix (g16_steps.h): window 17 starts one nent later
    // k_msm_gather's walk of a chunk (bp_layout.h: make_gather_steps): steps[2 t] = index of the first table entry of step t's window,
    // steps[2 t + 1] = (digit-word row of the step's scalar and window) << 1 | (window & 1); chunk c owns steps chunk_step0[c] .. chunk_step0[c + 1]
    const uint32_t* steps = nullptr; const uint32_t* chunk_step0 = nullptr;
};
ZKP_HD inline void msm_chunk_ref(const MsmView& m, uint32_t chunk, uint32_t row) {
    ge acc = ge_identity();
    uint32_t s = m.chunk_begin[chunk], w = m.chunk_win0[chunk];
    for (uint32_t left = m.chunk_nwin[chunk]; left > 0; left--) {
        const uint32_t srow = m.slot_scalar ? m.slot_scalar[s] : s;
        const int32_t d = ld_digit(m.digits, srow, w, row, m.rows);
        if (d != 0) acc = msm_accumulate_digit(acc, d, m.table + ((size_t)m.slot_base[s] * NWIN + w) * SUBTAB_W);
        if (++w == m.slot_nwin[s]) { s++; w = 0; }
    }
    st_ge(m.partial, chunk, row, m.rows, acc);
}
// sum the partial points of one target and encode.  thread = (target, row)
struct ReduceView {
    uint32_t rows, ntargets;
    const uint32_t* partial;
    const uint16_t* target_chunk_begin;  // [ntargets + 1]
    uint32_t* enc;                       // [ntargets][8][rows]
    const uint64_t* out_off;             // optional [rows]: also write target 0's 32 bytes to out + out_off[row]
    uint8_t* out;
    const uint32_t* corr;                // optional [ntargets][ACC_W]: point added to each target's sum (-(#chunks) * offset)
};
ZKP_HD inline void reduce_encode_thread(const ReduceView& r, uint32_t target, uint32_t row) {
    const uint32_t c0 = r.target_chunk_begin[target], c1 = r.target_chunk_begin[target + 1];
    ge acc = ld_ge(r.partial, c0, row, r.rows);
    for (uint32_t c = c0 + 1; c < c1; c++) acc = ge_add(acc, ld_ge(r.partial, c, row, r.rows));
    sc e; ge_ristretto_encode(e.v, acc);
    st_sc(r.enc, target, row, r.rows, e);
    if (r.out_off != nullptr && target == 0) put_bytes(r.out + r.out_off[row], e.v, 8);
}

// ------------------------------------------------------------------------------------------------
// transcript steps.  thread = job; `s` is the lane's STROBE image (LDS on the GPU).
ZKP_HD inline void strobe_save(const BpView& V, uint32_t job, const Strobe& s) {
    for (int i = 0; i < 50; i++) V.tstate[(size_t)i * V.M + job] = s.base[i * s.stride];
    V.tstate[(size_t)50 * V.M + job] = s.pos; V.tstate[(size_t)51 * V.M + job] = s.pos_begin;
}
ZKP_HD inline void strobe_load(const BpView& V, uint32_t job, Strobe& s) {
    for (int i = 0; i < 50; i++) s.base[i * s.stride] = V.tstate[(size_t)i * V.M + job];
    s.pos = V.tstate[(size_t)50 * V.M + job]; s.pos_begin = V.tstate[(size_t)51 * V.M + job];
}
ZKP_HD inline sc merlin_challenge_scalar(Strobe& s, const char* label, uint32_t label_len) {
    uint32_t w[16]; merlin_challenge_words(s, label, label_len, w, 16);
    return sc_from_wide(w);
}
ZKP_HD inline void merlin_append_scalar(Strobe& s, const char* label, uint32_t label_len, const sc& raw) { merlin_append_words(s, label, label_len, raw.v, 8); }

ZKP_HD inline void step_transcript1(const BpView& V, uint32_t job, Strobe& s) {
    const uint32_t M = V.M;
    switch (V.kind[job]) {   // Transcript::new(label), bulletproofs.rs:137,149,343,395,642
        case KIND_RANGE_MIN: merlin_init(s, "libzkp_range_min", 16); break;
        case KIND_RANGE_MAX: merlin_init(s, "libzkp_range_max", 16); break;
        case KIND_THRESHOLD: merlin_init(s, "libzkp_threshold", 16); break;
        case KIND_CONSISTENCY: merlin_init(s, "libzkp_consistency", 18); break;
        default: merlin_init(s, "libzkp_bulletproof", 18); break;
    }
    merlin_append_bytes(s, "dom-sep", 7, "rangeproof v1", 13);
    merlin_append_u64(s, "n", 1, V.n);
    merlin_append_u64(s, "m", 1, 1);
    const sc Ve = ld_sc(V.enc, 0, job, M), Ae = ld_sc(V.enc, 1, job, M), Se = ld_sc(V.enc, 2, job, M);
    merlin_append_words(s, "V", 1, Ve.v, 8);
    merlin_append_words(s, "A", 1, Ae.v, 8);
    merlin_append_words(s, "S", 1, Se.v, 8);
    const sc y = merlin_challenge_scalar(s, "y", 1);
    const sc z = merlin_challenge_scalar(s, "z", 1);
    strobe_save(V, job, s);
    st_sc(V.scal, SC_Y, job, M, y);
    st_sc(V.scal, SC_Z, job, M, z);
    // short power tables (28 products); step_poly forms y^i = y^(8a) y^b and y^-i per (i, job) lane with one product each
    const sc yinv = sc_invert(y);
    for (uint32_t h = 0; h < 2; h++) {
        const sc base = h ? yinv : y;
        sc p = sc_one();
        for (uint32_t b = 0; b < 8; b++) { st_sc(V.ypq, 16 * h + b, job, M, p); p = sc_mul(p, base); }
        sc q = sc_one();                                   // p = base^8 here
        for (uint32_t a = 0; a < 8; a++) { st_sc(V.ypq, 16 * h + 8 + a, job, M, q); q = sc_mul(q, p); }
    }
    uint8_t* pr = V.out + V.proof_off[job];
    put_bytes(pr, Ae.v, 8); put_bytes(pr + 32, Se.v, 8);
    put_bytes(V.out + V.commit_off[job], Ve.v, 8);
}

// polynomial coefficients.  thread = (i, job)
ZKP_HD inline void step_poly(const BpView& V, uint32_t i, uint32_t job) {
    const uint32_t M = V.M;
    const sc yi = sc_mul(ld_sc(V.ypq, 8 + (i >> 3), job, M), ld_sc(V.ypq, i & 7u, job, M));
    st_sc(V.yinvpow, i, job, M, sc_mul(ld_sc(V.ypq, 24 + (i >> 3), job, M), ld_sc(V.ypq, 16 + (i & 7u), job, M)));
    const sc z = ld_sc(V.scal, SC_Z, job, M);
    const sc zz = sc_mul(z, z);
    const uint32_t bit = (uint32_t)(V.v[job] >> i) & 1u;
    const sc one = sc_one();
    const sc l0 = bit ? sc_sub(one, z) : sc_neg(z);
    const sc arz = bit ? z : sc_sub(z, one);                       // a_R + z
    const sc two_i = sc_from_u64(1ull << i);
    const sc r0 = sc_add(sc_mul(yi, arz), sc_mul(zz, two_i));
    const sc l1 = ld_sc(V.tape, 2 + i, job, M);
    const sc r1 = sc_mul(yi, ld_sc(V.tape, 2 + V.n + i, job, M));
    st_sc(V.r0, i, job, M, r0); st_sc(V.r1, i, job, M, r1);
    st_sc(V.pp, i, job, M, sc_mul(l0, r0));
    st_sc(V.pp, 64 + i, job, M, sc_mul(sc_add(l0, l1), sc_add(r0, r1)));
    st_sc(V.pp, 128 + i, job, M, sc_mul(l1, r1));
}
// t0, t1, t2 = sums over i of the three product columns.  On the GPU eight lanes per job each add every 8th entry and a
// tree through LDS joins them (modular sums are exact, so the grouping does not change a bit); host emulation adds all
// eight parts in one thread.
struct ScTriple { sc a, b, c; };
ZKP_HD inline ScTriple triple_add(const ScTriple& x, const ScTriple& y) { return ScTriple{sc_add(x.a, y.a), sc_add(x.b, y.b), sc_add(x.c, y.c)}; }
ZKP_HD inline ScTriple step_poly_sum_part(const BpView& V, uint32_t part, uint32_t job) {
    const uint32_t M = V.M;
    ScTriple t{sc_zero(), sc_zero(), sc_zero()};
    for (uint32_t i = part; i < V.n; i += 8) {
        t.a = sc_add(t.a, ld_sc(V.pp, i, job, M));
        t.b = sc_add(t.b, ld_sc(V.pp, 64 + i, job, M));
        t.c = sc_add(t.c, ld_sc(V.pp, 128 + i, job, M));
    }
    return t;
}
ZKP_HD inline void step_poly_sum_finish(const BpView& V, uint32_t job, const ScTriple& t) {
    const uint32_t M = V.M;
    const sc t0 = t.a, t2 = t.c, t1 = sc_sub(sc_sub(t.b, t0), t2);
    st_sc(V.scal, SC_T0, job, M, t0); st_sc(V.scal, SC_T1, job, M, t1); st_sc(V.scal, SC_T2, job, M, t2);
    st_digits(V.d2, 0, job, M, t1, V.dig16);
    st_digits(V.d2, 2, job, M, t2, V.dig16);
}
ZKP_HD inline void step_poly_sum(const BpView& V, uint32_t job) {
    ScTriple t = step_poly_sum_part(V, 0, job);
    for (uint32_t p = 1; p < 8; p++) t = triple_add(t, step_poly_sum_part(V, p, job));
    step_poly_sum_finish(V, job, t);
}
// thread = job
ZKP_HD inline void step_transcript2(const BpView& V, uint32_t job, Strobe& s) {
    const uint32_t M = V.M;
    strobe_load(V, job, s);
    const sc T1e = ld_sc(V.enc, 0, job, M), T2e = ld_sc(V.enc, 1, job, M);
    merlin_append_words(s, "T_1", 3, T1e.v, 8);
    merlin_append_words(s, "T_2", 3, T2e.v, 8);
    const sc x = merlin_challenge_scalar(s, "x", 1);
    const sc xx = sc_mul(x, x);
    const sc z = ld_sc(V.scal, SC_Z, job, M), zz = sc_mul(z, z);
    const sc t0 = ld_sc(V.scal, SC_T0, job, M), t1 = ld_sc(V.scal, SC_T1, job, M), t2 = ld_sc(V.scal, SC_T2, job, M);
    const sc t_x = sc_add(sc_add(t0, sc_mul(t1, x)), sc_mul(t2, xx));
    const sc t1_bl = ld_sc(V.tape, 2 + 2 * V.n, job, M), t2_bl = ld_sc(V.tape, 3 + 2 * V.n, job, M);
    const sc t_x_bl = sc_add(sc_add(sc_mul(zz, ld_sc(V.gamma, 0, job, M)), sc_mul(t1_bl, x)), sc_mul(t2_bl, xx));
    const sc e_bl = sc_add(ld_sc(V.tape, 0, job, M), sc_mul(ld_sc(V.tape, 1, job, M), x));
    const sc r_tx = sc_to_raw(t_x), r_txb = sc_to_raw(t_x_bl), r_eb = sc_to_raw(e_bl);
    merlin_append_scalar(s, "t_x", 3, r_tx);
    merlin_append_scalar(s, "t_x_blinding", 12, r_txb);
    merlin_append_scalar(s, "e_blinding", 10, r_eb);
    const sc w = merlin_challenge_scalar(s, "w", 1);
    merlin_append_bytes(s, "dom-sep", 7, "ipp v1", 6);
    merlin_append_u64(s, "n", 1, V.n);
    strobe_save(V, job, s);
    st_sc(V.scal, SC_X, job, M, x); st_sc(V.scal, SC_W, job, M, w);
    uint8_t* pr = V.out + V.proof_off[job];
    put_bytes(pr + 64, T1e.v, 8); put_bytes(pr + 96, T2e.v, 8);
    put_bytes(pr + 128, r_tx.v, 8); put_bytes(pr + 160, r_txb.v, 8); put_bytes(pr + 192, r_eb.v, 8);
}
// l = l0 + l1*x, r = r0 + r1*x.  thread = (i, job)
ZKP_HD inline void step_lr_init(const BpView& V, uint32_t i, uint32_t job) {
    const uint32_t M = V.M;
    const sc z = ld_sc(V.scal, SC_Z, job, M), x = ld_sc(V.scal, SC_X, job, M);
    const uint32_t bit = (uint32_t)(V.v[job] >> i) & 1u;
    const sc l0 = bit ? sc_sub(sc_one(), z) : sc_neg(z);
    const sc a = sc_add(l0, sc_mul(ld_sc(V.tape, 2 + i, job, M), x));
    const sc b = sc_add(ld_sc(V.r0, i, job, M), sc_mul(ld_sc(V.r1, i, job, M), x));
    st_sc(V.ab, i, job, M, a); st_sc(V.ab, 64 + i, job, M, b);
}

// inner-product round r (k = n/2 >> r): fold a, b and the per-generator coefficients, emit MSM digits.
// thread = (i, job), i in [0, n)
ZKP_HD inline void step_round_prep(const BpView& V, uint32_t r, uint32_t i, uint32_t job) {
    const uint32_t M = V.M, p = V.lg - 1 - r, k = 1u << p, half = V.n >> 1;
    const uint32_t* abp = V.ab + (size_t)(r & 1) * (2 * 64 * 8) * M;
    uint32_t* abn = V.ab + (size_t)((r + 1) & 1) * (2 * 64 * 8) * M;
    sc u = sc_one(), uinv = sc_one();
    if (r > 0) { u = ld_sc(V.scal, SC_U, job, M); uinv = ld_sc(V.scal, SC_UINV, job, M); }
    // current (folded) vectors, length 2k, from the previous round's vectors of length 4k
    auto cur_a = [&](uint32_t x) -> sc {
        if (r == 0) return ld_sc(abp, x, job, M);
        return sc_add(sc_mul(ld_sc(abp, x, job, M), u), sc_mul(ld_sc(abp, x + 2 * k, job, M), uinv));
    };
    auto cur_b = [&](uint32_t x) -> sc {
        if (r == 0) return ld_sc(abp, 64 + x, job, M);
        return sc_add(sc_mul(ld_sc(abp, 64 + x, job, M), uinv), sc_mul(ld_sc(abp, 64 + x + 2 * k, job, M), u));
    };
    sc g, h;
    if (r == 0) { g = sc_one(); h = ld_sc(V.yinvpow, i, job, M); }
    else {
        const bool hi = (i >> (p + 1)) & 1u;
        g = sc_mul(ld_sc(V.gh, i, job, M), hi ? u : uinv);
        h = sc_mul(ld_sc(V.gh, 64 + i, job, M), hi ? uinv : u);
    }
    st_sc(V.gh, i, job, M, g); st_sc(V.gh, 64 + i, job, M, h);
    const uint32_t x = (i & (2 * k - 1)) ^ k;
    const sc ax = cur_a(x), bx = cur_b(x);
    if (i < 2 * k) { st_sc(abn, x, job, M, ax); st_sc(abn, 64 + x, job, M, bx); }
    const uint32_t bit = (i >> p) & 1u, rank = ((i >> (p + 1)) << p) | (i & (k - 1));
    st_digits(V.dr, bit ? 1 + rank : V.n + 2 + rank, job, M, sc_mul(ax, g), V.dig16);                    // L's G terms | R's G terms
    st_digits(V.dr, bit ? V.n + 2 + half + rank : 1 + half + rank, job, M, sc_mul(bx, h), V.dig16);      // R's H terms | L's H terms
    if (i < k) {   // here x = i + k
        const sc ai = cur_a(i), bi = cur_b(i);
        st_sc(V.pp, i, job, M, sc_mul(ai, bx));        // a_lo[i] * b_hi[i]
        st_sc(V.pp, 32 + i, job, M, sc_mul(ax, bi));   // a_hi[i] * b_lo[i]
    }
}
// thread = job
ZKP_HD inline void step_round_sum(const BpView& V, uint32_t r, uint32_t job) {
    const uint32_t M = V.M, k = (V.n >> 1) >> r;
    sc cL = sc_zero(), cR = sc_zero();
    for (uint32_t j = 0; j < k; j++) { cL = sc_add(cL, ld_sc(V.pp, j, job, M)); cR = sc_add(cR, ld_sc(V.pp, 32 + j, job, M)); }
    const sc w = ld_sc(V.scal, SC_W, job, M);
    st_digits(V.dr, 0, job, M, sc_mul(cL, w), V.dig16);
    st_digits(V.dr, V.n + 1, job, M, sc_mul(cR, w), V.dig16);
}
// thread = job
ZKP_HD inline void step_transcript_round(const BpView& V, uint32_t r, uint32_t job, Strobe& s) {
    const uint32_t M = V.M;
    strobe_load(V, job, s);
    const sc Le = ld_sc(V.enc, 0, job, M), Re = ld_sc(V.enc, 1, job, M);
    merlin_append_words(s, "L", 1, Le.v, 8);
    merlin_append_words(s, "R", 1, Re.v, 8);
    const sc u = merlin_challenge_scalar(s, "u", 1);
    const sc uinv = sc_invert(u);
    uint8_t* pr = V.out + V.proof_off[job];
    put_bytes(pr + 224 + 64 * r, Le.v, 8); put_bytes(pr + 256 + 64 * r, Re.v, 8);
    if (r + 1 < V.lg) {
        strobe_save(V, job, s);
        st_sc(V.scal, SC_U, job, M, u); st_sc(V.scal, SC_UINV, job, M, uinv);
    } else {
        const uint32_t* abp = V.ab + (size_t)(V.lg & 1u) * (2 * 64 * 8) * M;   // the buffer the last round wrote holds the length-2 vectors
        const sc a = sc_add(sc_mul(ld_sc(abp, 0, job, M), u), sc_mul(ld_sc(abp, 1, job, M), uinv));
        const sc b = sc_add(sc_mul(ld_sc(abp, 64, job, M), uinv), sc_mul(ld_sc(abp, 65, job, M), u));
        const sc ra = sc_to_raw(a), rb = sc_to_raw(b);
        put_bytes(pr + 224 + 64 * V.lg, ra.v, 8); put_bytes(pr + 256 + 64 * V.lg, rb.v, 8);
    }
}

// ------------------------------------------------------------------------------------------------
// Pedersen commitment tasks (value commitments of libzkp's framings).  thread = task
struct CtView {
    uint32_t C;
    const uint64_t* v; const uint32_t* seed_ix; const uint32_t* bl_ix; const uint32_t* seeds;
    uint32_t* digits;   // [2][DIGW][C]
    uint32_t dig16 = 0; // digit radix of the launch that consumes them (BpView::dig16)
};
ZKP_HD inline void step_ctask(const CtView& T, uint32_t c) {
    uint32_t seed[8]; const uint32_t* p = T.seeds + (size_t)T.seed_ix[c] * 8;
    ZKP_UNROLL for (int k = 0; k < 8; k++) seed[k] = p[k];
    const uint64_t v = T.v[c];
    st_digits_raw(T.digits, 0, c, T.C, sc_words((uint32_t)v, (uint32_t)(v >> 32), 0, 0, 0, 0, 0, 0), T.dig16);
    st_digits(T.digits, 1, c, T.C, tape_blinding(seed, T.bl_ix[c]), T.dig16);
}

// ------------------------------------------------------------------------------------------------
// Job construction for proof::range_proof::prove_range (/root/reference/src/proof/range_proof.rs:10-27 over
// bulletproofs.rs:112-178): op i -> jobs 2i (value-min, +blinding, "libzkp_range_min") and 2i+1
// (max-value, -blinding, "libzkp_range_max") plus one commitment task (value, blinding).  thread = op.
// Also writes every byte of the 1478-byte envelope that does not depend on the proofs
// (proof/mod.rs:23-36 header; bulletproofs.rs:160-177 body framing).
struct JobBuf {
    uint64_t* v; uint32_t* seed_ix; uint32_t* proof_ix; int32_t* bl_plus; int32_t* bl_minus; uint8_t* kind;
    uint64_t* proof_off; uint64_t* commit_off;
    uint64_t* ct_v; uint32_t* ct_seed_ix; uint32_t* ct_bl_ix; uint64_t* ct_off;
};
constexpr uint32_t RANGE_PROOF_BYTES = 1478, RP_BYTES = 672;
enum { ZKP_ST_OK = 0, ZKP_ST_INVALID_INPUT = 1 };
ZKP_HD inline void put_le(uint8_t* p, uint64_t x, int n) { for (int i = 0; i < n; i++) p[i] = (uint8_t)(x >> (8 * i)); }
ZKP_HD constexpr uint32_t range_body_bytes(uint32_t lg) { return 20 + 2 * (4 + rp_bytes(lg)) + 64; }   // bulletproofs.rs:160-177
ZKP_HD constexpr uint32_t range_envelope_bytes(uint32_t lg) { return 10 + range_body_bytes(lg) + 32; }
ZKP_HD inline void step_build_range(const JobBuf& J, uint32_t op, const uint64_t* value, const uint64_t* mn, const uint64_t* mx, uint32_t lg,
                                    uint8_t* out, uint64_t stride, uint32_t* out_len, int32_t* status) {
    const uint64_t val = value[op], lo = mn[op], hi = mx[op];
    const uint64_t max_diff = lg >= 6 ? ~0ull : (1ull << (1u << lg)) - 1;      // bulletproofs.rs:94-100,121-129
    const bool ok = lo <= hi && val >= lo && val <= hi                  // validation.rs:5-18
                    && val - lo <= max_diff && hi - val <= max_diff;
    const uint32_t RP_BYTES = rp_bytes(lg), body = range_body_bytes(lg);
    status[op] = ok ? ZKP_ST_OK : ZKP_ST_INVALID_INPUT;
    out_len[op] = ok ? range_envelope_bytes(lg) : 0;
    const uint64_t base = (uint64_t)op * stride;
    for (uint32_t j = 0; j < 2; j++) {
        const uint32_t job = 2 * op + j;
        J.v[job] = ok ? (j == 0 ? val - lo : hi - val) : 0;
        J.seed_ix[job] = op; J.proof_ix[job] = j;
        J.bl_plus[job] = j == 0 ? 0 : -1; J.bl_minus[job] = j == 0 ? -1 : 0;
        J.kind[job] = (uint8_t)(j == 0 ? KIND_RANGE_MIN : KIND_RANGE_MAX);
        J.proof_off[job] = base + 34 + (uint64_t)j * (4 + RP_BYTES);
        J.commit_off[job] = base + 10 + 20 + 2 * (4 + RP_BYTES) + 32 * j;
    }
    J.ct_v[op] = ok ? val : 0; J.ct_seed_ix[op] = op; J.ct_bl_ix[op] = 0;
    J.ct_off[op] = base + 10 + body;
    uint8_t* o = out + base;
    o[0] = 2; o[1] = 1; put_le(o + 2, body, 4); put_le(o + 6, 32, 4);
    put_le(o + 10, lo, 8); put_le(o + 18, hi, 8); put_le(o + 26, 1u << lg, 4);
    put_le(o + 30, RP_BYTES, 4); put_le(o + 34 + RP_BYTES, RP_BYTES, 4);
}

}  // namespace zkp
