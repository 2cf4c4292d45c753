// Batched verification of libzkp range proofs: proof::range_proof::verify_range
// (/root/reference/src/proof/range_proof.rs:28-47) -> BulletproofsBackend::verify_range_with_bounds
// (/root/reference/src/backend/bulletproofs.rs:181-295) -> two `RangeProof::verify_single` (crate bulletproofs ^5.0).
// Per-thread steps shared by the kernels (bpv_kernels.hip) and the host emulation used by the no-GPU tests.
//
// One envelope = two jobs (value-min under "libzkp_range_min", max-value under "libzkp_range_max").  Per job the two
// verification equations of verify_single are folded into ONE multiscalar check with a transcript-derived weight c
// (upstream does the same with a random c):
//     A + x S - e_bl B~ + w (t_x - a b) B + sum u_j^2 L_j + u_j^-2 R_j + sum -(z + a s_i) G_i
//       + sum (z + y^-i (z^2 2^i - b s_{63-i})) H_i  +  c [ (t_x - delta) B + t_xb B~ - z^2 V - x T_1 - x^2 T_2 ]  ==  0
// The 130 generator terms go through the fixed-base MSM kernel of the prover (same tables); the 17 proof points are
// multiplied one lane each (signed radix-4, two doublings per digit) and land in extra chunk slots of the same partial
// array, so the prover's partial-sum and encode kernels finish the job: the sum is the identity iff its ristretto
// encoding is 32 zero bytes.
#pragma once
#include "bp_steps.h"
#include "edg.h"
#include "sha256_dev.h"

namespace zkp {

enum { VP_V = 0, VP_A, VP_S, VP_T1, VP_T2, VP_L = 5, VP_R = 11, VP_NUM = 17 };
enum { VS_Z = 0, VS_ZZ, VS_YINV, VS_A, VS_B, VS_U = 5, VS_UINV = 11, VS_NUM = 17 };
constexpr int32_t VFY_UNSUPPORTED = 2;      // reserved verdict (every bit width the reference accepts -- 8, 16, 32, 64 -- is verified)

// Batch check (bpv_kernels.hip: k_rlc_*): the 17 M weighted proof-point terms go through ONE bucket-method MSM with signed
// radix-2048 digits: 24 windows of 1024 buckets.
constexpr uint32_t RLC_MIN_JOBS = 4096;      // measured break-even ~2100 jobs (profiles/r02_verify_rates.json); ZKP_HIP_BATCH_VERIFY_MIN overrides
// Window width 11: 23 windows cover the 253 bits of a scalar < l ~ 2^252 exactly, so every window's digits are uniform over the
// 1024 bucket magnitudes (a width that leaves a short top window puts half of all points into one or two buckets there); the
// 24th window only ever holds the recoding carry.
constexpr uint32_t RLC_WBITS = 11, RLC_NWIN = 24, RLC_NBUCKET = 1u << (RLC_WBITS - 1), RLC_SEG = 32, RLC_NSEG = RLC_NBUCKET / RLC_SEG, RLC_NIELS_W = 32;
struct RlcView {
    uint32_t N;                   // 17 * M point terms, index = p * M + job
    uint32_t* niels;              // [N][32] affine-Niels form of the decoded points (30 words used)
    int16_t* dig;                 // [NWIN][N] signed digits of the weighted scalars
    uint32_t* count;              // [NWIN][NBUCKET + 1] points per bucket magnitude (index 0 unused)
    uint32_t* start;              // [NWIN][NBUCKET + 2] exclusive prefix sums
    uint32_t* cursor;             // [NWIN][NBUCKET + 1] scatter cursors
    uint32_t* sorted;             // [NWIN][N] point index | sign << 31, grouped by bucket
    uint32_t* bucket;             // [NWIN * NBUCKET][40] bucket sums (extended coordinates, word-major per point)
    uint32_t* seg;                // [NWIN * NSEG][40]
    const uint32_t* fixed_sum;    // [40] the generator part (one-row fixed-base MSM of the summed coefficients)
    uint32_t* result;             // [8] ristretto encoding of the whole combination, [8] = 1 iff it is the identity
};

struct VfyView {
    uint32_t M;                   // jobs
    const uint8_t* in;            // the envelopes
    uint64_t* proof_off;          // [M] byte offset of the job's 672-byte RangeProof
    uint64_t* venc_off;           // [M] byte offset of the job's 32-byte value commitment encoding
    uint8_t* kind;                // [M] transcript label
    uint8_t* lgn;                 // [M] log2 of the job's bit width (read from the envelope: bulletproofs.rs:211-216,567-572)
    int32_t* bad;                 // [M] nonzero: reject (1) / unsupported (2)
    uint32_t* pts;                // [17][40][M] decoded proof points
    uint32_t* scal;               // [VS_NUM][8][M] Montgomery-form scalars shared between steps
    uint32_t* digits;             // [130][DIGW][M] fixed-base digits (zero-initialised: rejected jobs contribute nothing)
    uint32_t* vscal;              // [17][8][M] raw scalars of the proof points
    uint32_t* partial;            // the MSM partial array; proof-point products go to chunks var_chunk0 + p
    uint32_t var_chunk0;
    const uint32_t* table;        // generator tables (parse step: min*B, max*B): radix 2^16 in HBM (edg.h) when dig16, else the radix-1024 tables
    uint32_t dig16 = 0;           // radix of `table` and of the fixed-base digit rows (1: the device; 0: the host emulation's small tables)
    // batch mode (random linear combination over the whole batch): rho = [8][M] per-job weights (Montgomery scalars), or
    // null for the per-job check.  With weights every coefficient is multiplied by rho_job; the generator coefficients then go to
    // fterm ([130][8][M] Montgomery scalars, summed over the jobs afterwards) instead of per-job digit rows.
    const uint32_t* rho;
    uint32_t* fterm;
    // consistency proofs: a variable number of jobs per envelope
    const uint32_t* job_base;     // [n + 1] first job of each envelope (host: k - 1 jobs where the framing can hold k commitments)
    int32_t* env_bad;             // [n] envelope-level verdict of the framing / commitment checks
};

ZKP_HD inline void ld_bytes_words(uint32_t* w, const uint8_t* p, uint32_t nwords) {
    for (uint32_t i = 0; i < nwords; i++) w[i] = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
}
ZKP_HD inline uint32_t ld_u32(const uint8_t* p) { uint32_t w; ld_bytes_words(&w, p, 1); return w; }
ZKP_HD inline uint64_t ld_u64(const uint8_t* p) { return (uint64_t)ld_u32(p) | ((uint64_t)ld_u32(p + 4) << 32); }

// RFC 9496 4.3.1: false for non-canonical, negative or off-curve encodings
ZKP_HD inline bool ge_ristretto_decode(ge& p, const uint32_t w[8]) {
    const fe s = fe_fromwords(w);
    uint32_t chk[8]; fe_towords(chk, s);
    bool canonical = (w[7] >> 31) == 0;
    for (int k = 0; k < 8; k++) canonical = canonical && chk[k] == w[k];
    const fe ss = fe_sq(s);
    const fe u1 = fe_carry(fe_sub(fe_one(), ss)), u2 = fe_carry(fe_add(fe_one(), ss));
    const fe u2s = fe_sq(u2);
    const fe v = fe_carry(fe_sub(fe_carry(fe_neg(fe_mul(fe_sq(u1), fe_const_d()))), u2s));
    fe invsqrt;
    const bool was_square = fe_sqrt_ratio_m1(invsqrt, fe_one(), fe_mul(v, u2s));
    const fe den_x = fe_mul(invsqrt, u2);
    const fe den_y = fe_mul(fe_mul(invsqrt, den_x), v);
    const fe t = fe_mul(s, den_x);
    p.X = fe_abs(fe_carry(fe_add(t, t)));
    p.Y = fe_mul(u1, den_y);
    p.Z = fe_one();
    p.T = fe_mul(p.X, p.Y);
    return canonical && (w[0] & 1u) == 0 && was_square && !fe_isneg(p.T) && !fe_iszero(p.Y);
}
ZKP_HD inline bool words_are_zero(const uint32_t w[8]) { uint32_t o = 0; for (int k = 0; k < 8; k++) o |= w[k]; return o == 0; }
// Scalar::from_canonical_bytes: raw < l
ZKP_HD inline bool sc_raw_is_canonical(const sc& raw) {
    const uint32_t L[8] = {0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u};
    for (int i = 7; i >= 0; i--) { if (raw.v[i] < L[i]) return true; if (raw.v[i] > L[i]) return false; }
    return false;
}
// k * P for a raw scalar (< 2^253), signed radix-4 digits {-1, 0, 1, 2}, MSB first: 2 doublings + at most 1 addition per digit
ZKP_HD inline ge ge_scalarmult_raw(const ge& P, const sc& k) {
    const ge P2 = ge_dbl(P);
    // recode into 127 digits (2 bits each) with carry: 3 -> -1 carry 1
    uint32_t dig[8];                                   // 16 two-bit fields per word: 0, 1, 2, or 3 meaning -1
    uint32_t carry = 0;
    for (int w = 0; w < 8; w++) {
        uint32_t out = 0;
        for (int j = 0; j < 16; j++) {
            uint32_t d = ((k.v[w] >> (2 * j)) & 3u) + carry;      // 0..4
            carry = d >= 3u ? 1u : 0u;                           // 3 -> -1, 4 -> 0, both with carry
            out |= (d & 3u) << (2 * j);
        }
        dig[w] = out;
    }
    // k < 2^253: the top digits (bits 254..255) are zero and absorb the last carry
    ge acc = ge_identity();
    for (int i = 127; i >= 0; i--) {
        acc = ge_dbl(ge_dbl(acc));
        const uint32_t d = (dig[i >> 4] >> (2 * (i & 15))) & 3u;
        if (d != 0) {
            ge q = d == 2 ? P2 : P;
            if (d == 3) q = ge_neg(P);
            acc = ge_add(acc, q);
        }
    }
    return acc;
}

// v * B for a 64-bit v through the window tables of the basepoint (either radix: VfyView::dig16)
ZKP_HD inline ge vfy_mul_b_u64(const VfyView& V, uint64_t v) {
    ge acc = ge_identity();
    const sc raw = sc_words((uint32_t)v, (uint32_t)(v >> 32), 0, 0, 0, 0, 0, 0);
    if (V.dig16) {
        uint32_t d[8]; sc_recode_signed65536(d, raw);
        for (uint32_t win = 0; win < EDG_NWIN_U64; win++) {
            const int32_t a = (int32_t)(int16_t)(d[win >> 1] >> (16 * (win & 1)));
            if (a != 0) acc = edg_accumulate_from(acc, a, V.table, BASE_B, win);
        }
        return acc;
    }
    uint32_t d[DIGW]; sc_recode_signed1024(d, raw);
    for (uint32_t win = 0; win < NWIN_U64; win++) {
        const int32_t a = (int32_t)(int16_t)(d[win >> 1] >> (16 * (win & 1)));
        if (a != 0) acc = msm_accumulate_digit(acc, a, V.table + ((size_t)BASE_B * NWIN + win) * SUBTAB_W);
    }
    return acc;
}

// ---- step 0: envelope framing (bulletproofs.rs:181-295 through proof_helpers.rs:12-36).  thread = envelope
ZKP_HD inline void step_vparse(const VfyView& V, uint32_t i, const uint8_t* env, uint64_t env_off, uint32_t len, uint64_t mn, uint64_t mx) {
    const uint32_t j0 = 2 * i, j1 = 2 * i + 1;
    V.kind[j0] = KIND_RANGE_MIN; V.kind[j1] = KIND_RANGE_MAX; V.lgn[j0] = V.lgn[j1] = 6;
    V.proof_off[j0] = V.proof_off[j1] = 0; V.venc_off[j0] = V.venc_off[j1] = 0;
    V.bad[j0] = V.bad[j1] = 1;
    if (mn > mx || len < 10 || len > 1024u * 1024u || env[0] != 2 || env[1] != 1) return;
    const uint32_t bl = ld_u32(env + 2), cl = ld_u32(env + 6);
    if (bl > 900u * 1024u || cl != 32 || (uint64_t)10 + bl + cl != len) return;
    const uint8_t* body = env + 10; const uint8_t* comm = env + 10 + bl;
    uint32_t w[8];
    ge vc; ld_bytes_words(w, comm, 8); if (!ge_ristretto_decode(vc, w)) return;
    if (bl < 20 || ld_u64(body) != mn || ld_u64(body + 8) != mx) return;
    const uint32_t n_bits = ld_u32(body + 16);
    uint32_t pos = 20, rl[2], rp[2];
    for (int k = 0; k < 2; k++) {
        if (bl - pos < 4) return;
        rl[k] = ld_u32(body + pos); pos += 4;
        if (bl - pos < rl[k]) return;
        rp[k] = pos; pos += rl[k];
    }
    if (bl - pos < 64) return;
    ge cm, cx; uint32_t wm[8], wx[8];
    ld_bytes_words(wm, body + pos, 8); ld_bytes_words(wx, body + pos + 32, 8);
    if (!ge_ristretto_decode(cm, wm) || !ge_ristretto_decode(cx, wx)) return;
    if (!(n_bits == 8 || n_bits == 16 || n_bits == 32 || n_bits == 64)) return;
    // commitments of the two sub-proofs must be C - min*B and max*B - C
    const ge mb = vfy_mul_b_u64(V, mn), xb = vfy_mul_b_u64(V, mx);
    uint32_t e1[8], e2[8];
    ge_ristretto_encode(e1, ge_add(vc, ge_neg(mb)));
    ge_ristretto_encode(e2, ge_add(xb, ge_neg(vc)));
    for (int k = 0; k < 8; k++) if (e1[k] != wm[k] || e2[k] != wx[k]) return;
    uint32_t lg = 3; while ((1u << lg) != n_bits) lg++;
    if (rl[0] != rp_bytes(lg) || rl[1] != rp_bytes(lg)) return;
    V.lgn[j0] = V.lgn[j1] = (uint8_t)lg;
    V.proof_off[j0] = env_off + 10 + rp[0]; V.proof_off[j1] = env_off + 10 + rp[1];
    V.venc_off[j0] = env_off + 10 + pos; V.venc_off[j1] = env_off + 10 + pos + 32;
    V.bad[j0] = V.bad[j1] = 0;
}

// ---- step 0 for threshold proofs (threshold_proof.rs:34-47 + bulletproofs.rs:550-626): one job per envelope, the
// sub-proof's commitment must be C - threshold*B.  thread = envelope
ZKP_HD inline void step_vparse_threshold(const VfyView& V, uint32_t i, const uint8_t* env, uint64_t env_off, uint32_t len, uint64_t threshold) {
    V.kind[i] = KIND_THRESHOLD; V.lgn[i] = 6; V.proof_off[i] = 0; V.venc_off[i] = 0; V.bad[i] = 1;
    if (len < 10 || len > 1024u * 1024u || env[0] != 2 || env[1] != 3) return;
    const uint32_t bl = ld_u32(env + 2), cl = ld_u32(env + 6);
    if (bl > 900u * 1024u || cl != 32 || (uint64_t)10 + bl + cl != len) return;
    const uint8_t* body = env + 10; const uint8_t* comm = env + 10 + bl;
    if (bl < 12 || ld_u64(body) != threshold) return;
    const uint32_t n_bits = ld_u32(body + 8);
    if (bl < 16) return;
    const uint32_t rl = ld_u32(body + 12);
    if (bl - 16 < rl || bl - 16 - rl < 32) return;
    if (!(n_bits == 8 || n_bits == 16 || n_bits == 32 || n_bits == 64)) return;
    const uint8_t* dc = body + 16 + rl;
    ge d, sp; uint32_t wd[8], wsp[8];
    ld_bytes_words(wd, dc, 8); ld_bytes_words(wsp, comm, 8);
    if (!ge_ristretto_decode(d, wd) || !ge_ristretto_decode(sp, wsp)) return;
    const ge tb = vfy_mul_b_u64(V, threshold);
    uint32_t e[8];
    ge_ristretto_encode(e, ge_add(sp, ge_neg(tb)));
    for (int k = 0; k < 8; k++) if (e[k] != wd[k]) return;
    uint32_t lg = 3; while ((1u << lg) != n_bits) lg++;
    if (rl != rp_bytes(lg)) return;
    V.lgn[i] = (uint8_t)lg;
    V.proof_off[i] = env_off + 10 + 16; V.venc_off[i] = env_off + 10 + 16 + rl; V.bad[i] = 0;
}

// ---- step 0 for consistency proofs (consistency_proof.rs:24-32 + bulletproofs.rs:439-547): k commitments, k - 1 range
// proofs of the successive differences, SHA-256 of the commitments as the envelope commitment.  thread = envelope
ZKP_HD inline void step_vparse_consistency(const VfyView& V, uint32_t e, const uint8_t* env, uint64_t env_off, uint32_t len) {
    const uint32_t jb = V.job_base[e], je = V.job_base[e + 1];
    for (uint32_t j = jb; j < je; j++) { V.kind[j] = KIND_CONSISTENCY; V.lgn[j] = 6; V.proof_off[j] = 0; V.venc_off[j] = 0; V.bad[j] = 1; }
    V.env_bad[e] = 1;
    if (len < 10 || len > 1024u * 1024u || env[0] != 2 || env[1] != 6) return;
    const uint32_t bl = ld_u32(env + 2), cl = ld_u32(env + 6);
    if (bl > 900u * 1024u || cl != 32 || (uint64_t)10 + bl + cl != len) return;
    const uint8_t* body = env + 10; const uint8_t* comm = env + 10 + bl;
    if (bl < 4) return;
    const uint32_t k = ld_u32(body);
    uint64_t left = bl - 4, pos = 4;
    if (k == 0 || left < 32ull * k || k - 1 != je - jb) return;
    const uint8_t* commits = body + 4; pos += 32ull * k; left -= 32ull * k;
    uint8_t dg[32]; sha256_bytes(dg, commits, 32ull * k);
    for (int i = 0; i < 32; i++) if (dg[i] != comm[i]) return;
    uint32_t w[8];
    for (uint32_t i = 0; i < k; i++) { ge p; ld_bytes_words(w, commits + 32 * i, 8); if (!ge_ristretto_decode(p, w)) return; }
    bool lens_ok = true;
    for (uint32_t i = 1; i < k; i++) {
        if (left < 4) return;
        const uint32_t rl = ld_u32(body + pos); pos += 4; left -= 4;
        if (left < rl) return;
        V.proof_off[jb + i - 1] = env_off + 10 + pos; pos += rl; left -= rl;
        lens_ok = lens_ok && rl == RP_BYTES;
    }
    ge prev; ld_bytes_words(w, commits, 8); (void)ge_ristretto_decode(prev, w);
    for (uint32_t i = 1; i < k; i++) {
        if (left < 32) return;
        ge d, cur; uint32_t wd[8], e2[8];
        ld_bytes_words(wd, body + pos, 8);
        if (!ge_ristretto_decode(d, wd)) return;
        ld_bytes_words(w, commits + 32 * i, 8); (void)ge_ristretto_decode(cur, w);
        ge_ristretto_encode(e2, ge_add(cur, ge_neg(prev)));
        for (int q = 0; q < 8; q++) if (e2[q] != wd[q]) return;
        if (!lens_ok) return;                                   // verify_single rejects a proof of the wrong length (after these checks, as upstream orders them)
        V.venc_off[jb + i - 1] = env_off + 10 + pos;
        prev = cur; pos += 32; left -= 32;
    }
    V.env_bad[e] = 0;
    for (uint32_t j = jb; j < je; j++) V.bad[j] = 0;
}
// verdict for a variable number of jobs per envelope
ZKP_HD inline void step_vfinal_ranges(const VfyView& V, const uint32_t* enc, uint32_t e, uint8_t* ok) {
    uint32_t verdict = V.env_bad[e] ? 0u : 1u;
    for (uint32_t j = V.job_base[e]; j < V.job_base[e + 1]; j++) {
        uint32_t o = 0; for (int q = 0; q < 8; q++) o |= enc[(size_t)q * V.M + j];
        if (V.bad[j] || o != 0) verdict = 0;
    }
    ok[e] = (uint8_t)verdict;
}

// ---- step 1: decode the 5 + 2 lg n points of a job.  thread = (p, job)
ZKP_HD inline bool vpoint_present(uint32_t p, uint32_t lg) { return p < VP_L || (p < VP_R ? p - VP_L : p - VP_R) < lg; }
ZKP_HD inline void step_vdecode(const VfyView& V, uint32_t p, uint32_t job) {
    if (V.bad[job] || !vpoint_present(p, V.lgn[job])) return;
    const uint8_t* pr = V.in + V.proof_off[job];
    const uint8_t* src = p == VP_V ? V.in + V.venc_off[job] : p < VP_L ? pr + 32 * (p - VP_A) : p < VP_R ? pr + 224 + 64 * (p - VP_L) : pr + 224 + 64 * (p - VP_R) + 32;
    uint32_t w[8]; ld_bytes_words(w, src, 8);
    ge pt;
    const bool ok = ge_ristretto_decode(pt, w);
    if (!ok || (p != VP_V && words_are_zero(w))) { V.bad[job] = 1; return; }     // A, S, T1, T2, L_j, R_j must not be the identity
    st_ge(V.pts, p, job, V.M, pt);
}

// coefficient of generator `base` for this job: per-job digits, or the weighted term of the batch check
ZKP_HD inline void st_gen_coef(const VfyView& V, uint32_t base, uint32_t job, const sc& coef) {
    if (V.rho == nullptr) st_digits(V.digits, base, job, V.M, coef, V.dig16);
    else st_sc(V.fterm, base, job, V.M, sc_mul(coef, ld_sc(V.rho, 0, job, V.M)));
}
// raw scalar of proof point p (weighted in batch mode)
ZKP_HD inline void st_point_scalar(const VfyView& V, uint32_t p, uint32_t job, const sc& mont) {
    st_sc(V.vscal, p, job, V.M, sc_to_raw(V.rho == nullptr ? mont : sc_mul(mont, ld_sc(V.rho, 0, job, V.M))));
}

// ---- step 2: transcript replay, job-level scalars.  thread = job
ZKP_HD inline void step_vtranscript(const VfyView& V, uint32_t job, Strobe& s) {
    if (V.bad[job]) return;
    const uint32_t M = V.M, lg = V.lgn[job], n = 1u << lg;
    const uint8_t* pr = V.in + V.proof_off[job];
    sc r_tx, r_txb, r_eb, r_a, r_b;
    ld_bytes_words(r_tx.v, pr + 128, 8); ld_bytes_words(r_txb.v, pr + 160, 8); ld_bytes_words(r_eb.v, pr + 192, 8);
    ld_bytes_words(r_a.v, pr + 224 + 64 * lg, 8); ld_bytes_words(r_b.v, pr + 256 + 64 * lg, 8);
    if (!sc_raw_is_canonical(r_tx) || !sc_raw_is_canonical(r_txb) || !sc_raw_is_canonical(r_eb) || !sc_raw_is_canonical(r_a) || !sc_raw_is_canonical(r_b)) { V.bad[job] = 1; return; }
    switch (V.kind[job]) {
        case KIND_RANGE_MIN: merlin_init(s, "libzkp_range_min", 16); break;
        case KIND_RANGE_MAX: merlin_init(s, "libzkp_range_max", 16); break;
        case KIND_CONSISTENCY: merlin_init(s, "libzkp_consistency", 18); break;
        default: merlin_init(s, "libzkp_threshold", 16); break;
    }
    merlin_append_bytes(s, "dom-sep", 7, "rangeproof v1", 13);
    merlin_append_u64(s, "n", 1, n);
    merlin_append_u64(s, "m", 1, 1);
    uint32_t w[8];
    ld_bytes_words(w, V.in + V.venc_off[job], 8); merlin_append_words(s, "V", 1, w, 8);
    ld_bytes_words(w, pr, 8); merlin_append_words(s, "A", 1, w, 8);
    ld_bytes_words(w, pr + 32, 8); merlin_append_words(s, "S", 1, w, 8);
    const sc y = merlin_challenge_scalar(s, "y", 1);
    const sc z = merlin_challenge_scalar(s, "z", 1);
    ld_bytes_words(w, pr + 64, 8); merlin_append_words(s, "T_1", 3, w, 8);
    ld_bytes_words(w, pr + 96, 8); merlin_append_words(s, "T_2", 3, w, 8);
    const sc x = merlin_challenge_scalar(s, "x", 1);
    merlin_append_scalar(s, "t_x", 3, r_tx);
    merlin_append_scalar(s, "t_x_blinding", 12, r_txb);
    merlin_append_scalar(s, "e_blinding", 10, r_eb);
    const sc wch = merlin_challenge_scalar(s, "w", 1);
    merlin_append_bytes(s, "dom-sep", 7, "ipp v1", 6);
    merlin_append_u64(s, "n", 1, n);
    sc u[6];
    for (uint32_t j = 0; j < lg; j++) {
        ld_bytes_words(w, pr + 224 + 64 * j, 8); merlin_append_words(s, "L", 1, w, 8);
        ld_bytes_words(w, pr + 224 + 64 * j + 32, 8); merlin_append_words(s, "R", 1, w, 8);
        u[j] = merlin_challenge_scalar(s, "u", 1);
    }
    // The weight that folds the two verification equations (upstream draws it from an RNG seeded outside the proof) is derived
    // only after EVERYTHING the prover sent has been absorbed -- including the inner-product argument's final scalars a and b,
    // which the proof transcript itself never absorbs -- so no part of the proof can be chosen with knowledge of it.
    merlin_append_scalar(s, "a", 1, r_a);
    merlin_append_scalar(s, "b", 1, r_b);
    const sc c = merlin_challenge_scalar(s, "libzkp-amd batch weight", 23);   // verifier-internal: not part of the proof
    // one inversion for y, u_0..u_{lg-1} (Montgomery's trick)
    sc pre[7]; sc run = y;
    pre[0] = sc_one();
    for (uint32_t j = 0; j < lg; j++) { pre[j + 1] = run; run = sc_mul(run, u[j]); }
    sc inv = sc_invert(run);
    sc uinv[6];
    for (int j = (int)lg - 1; j >= 0; j--) { uinv[j] = sc_mul(inv, pre[j + 1]); inv = sc_mul(inv, u[j]); }
    const sc yinv = inv;
    const sc t_x = sc_from_raw256(r_tx), t_xb = sc_from_raw256(r_txb), e_bl = sc_from_raw256(r_eb), a = sc_from_raw256(r_a), b = sc_from_raw256(r_b);
    const sc zz = sc_mul(z, z), xx = sc_mul(x, x);
    sc sum_y = sc_zero(), yp = sc_one();
    for (uint32_t i = 0; i < n; i++) { sum_y = sc_add(sum_y, yp); yp = sc_mul(yp, y); }
    const sc delta = sc_sub(sc_mul(sc_sub(z, zz), sum_y), sc_mul(sc_mul(zz, z), sc_from_u64(lg >= 6 ? ~0ull : (1ull << n) - 1)));   // <1, 2^n> = 2^n - 1
    st_sc(V.scal, VS_Z, job, M, z); st_sc(V.scal, VS_ZZ, job, M, zz); st_sc(V.scal, VS_YINV, job, M, yinv);
    st_sc(V.scal, VS_A, job, M, a); st_sc(V.scal, VS_B, job, M, b);
    for (uint32_t j = 0; j < lg; j++) { st_sc(V.scal, VS_U + j, job, M, u[j]); st_sc(V.scal, VS_UINV + j, job, M, uinv[j]); }
    // generator coefficients of B and B~
    st_gen_coef(V, BASE_B, job, sc_add(sc_mul(wch, sc_sub(t_x, sc_mul(a, b))), sc_mul(c, sc_sub(t_x, delta))));
    st_gen_coef(V, BASE_BB, job, sc_sub(sc_mul(c, t_xb), e_bl));
    // proof-point scalars
    const sc cz = sc_mul(c, zz);
    st_point_scalar(V, VP_V, job, sc_neg(cz));
    st_point_scalar(V, VP_A, job, sc_one());
    st_point_scalar(V, VP_S, job, x);
    st_point_scalar(V, VP_T1, job, sc_neg(sc_mul(c, x)));
    st_point_scalar(V, VP_T2, job, sc_neg(sc_mul(c, xx)));
    for (uint32_t j = 0; j < lg; j++) {
        st_point_scalar(V, VP_L + j, job, sc_mul(u[j], u[j]));
        st_point_scalar(V, VP_R + j, job, sc_mul(uinv[j], uinv[j]));
    }
}

// ---- step 3: coefficients of G_i and H_i.  thread = (i, job), i in [0, 64): generators past the job's width keep zero digits
ZKP_HD inline void step_vscalars(const VfyView& V, uint32_t i, uint32_t job) {
    if (V.bad[job]) return;
    const uint32_t M = V.M, lg = V.lgn[job];
    if (i >> lg) return;
    sc s_i = sc_one(), s_r = sc_one();                 // s_i and s_{n-1-i}
    for (uint32_t j = 0; j < lg; j++) {
        const sc u = ld_sc(V.scal, VS_U + j, job, M), ui = ld_sc(V.scal, VS_UINV + j, job, M);
        const bool bit = (i >> (lg - 1 - j)) & 1u;
        s_i = sc_mul(s_i, bit ? u : ui);
        s_r = sc_mul(s_r, bit ? ui : u);
    }
    const sc z = ld_sc(V.scal, VS_Z, job, M), zz = ld_sc(V.scal, VS_ZZ, job, M), a = ld_sc(V.scal, VS_A, job, M), b = ld_sc(V.scal, VS_B, job, M);
    sc yip = sc_one(), base = ld_sc(V.scal, VS_YINV, job, M);          // y^-i by square and multiply over the 6 bits of i
    for (uint32_t k = 0; k < 6; k++) { if ((i >> k) & 1u) yip = sc_mul(yip, base); base = sc_mul(base, base); }
    st_gen_coef(V, BASE_G + i, job, sc_neg(sc_add(z, sc_mul(a, s_i))));
    const sc h = sc_add(z, sc_mul(yip, sc_sub(sc_mul(zz, sc_from_u64(1ull << i)), sc_mul(b, s_r))));
    st_gen_coef(V, BASE_H + i, job, h);
}

// ---- step 4: scalar * proof point.  thread = (p, job)
ZKP_HD inline void step_vvarbase(const VfyView& V, uint32_t p, uint32_t job) {
    const uint32_t M = V.M;
    ge r = ge_identity();
    if (!V.bad[job] && vpoint_present(p, V.lgn[job])) {
        const ge pt = ld_ge(V.pts, p, job, M);
        r = (p == VP_A && V.rho == nullptr) ? pt : ge_scalarmult_raw(pt, ld_sc(V.vscal, p, job, M));
    }
    st_ge(V.partial, V.var_chunk0 + p, job, M, r);
}

// ---- step 5: verdict.  thread = envelope (jobs_per = 2 for range, 1 for threshold); enc = [1][8][M] encodings of the per-job sums
ZKP_HD inline void step_vfinal(const VfyView& V, const uint32_t* enc, uint32_t i, uint8_t* ok, uint32_t jobs_per) {
    uint32_t verdict = 1;
    for (uint32_t j = jobs_per * i; j < jobs_per * (i + 1); j++) {
        if (V.bad[j] == VFY_UNSUPPORTED) { verdict = VFY_UNSUPPORTED; break; }
        uint32_t o = 0; for (int k = 0; k < 8; k++) o |= enc[(size_t)k * V.M + j];
        if (V.bad[j] || o != 0) verdict = 0;
    }
    ok[i] = (uint8_t)verdict;
}

}  // namespace zkp
