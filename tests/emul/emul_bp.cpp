// TEST INFRASTRUCTURE: runs the per-thread step functions of the HIP prover on the CPU (plain loops in
// place of kernel launches) so the no-GPU test tier can compare them with the oracle byte for byte.
// Not part of the product library; the product has no host proving path.
#include "../../libzkp_amd/csrc/bp_layout.h"
#include "../../libzkp_amd/csrc/bp_verify.h"
#include "../../libzkp_amd/csrc/edg.h"
#include <vector>
#include <cstdlib>
#include <cstring>
using namespace zkp;

static std::vector<uint32_t> g_table;
static void ensure_table() {
    if (!g_table.empty()) return;
    g_table.resize((size_t)NBASE * NWIN * SUBTAB_W);
    ge gens[NBASE]; host_generators(gens);
    for (uint32_t b = 0; b < NBASE; b++) host_build_table_for_base(g_table.data() + (size_t)b * NWIN * SUBTAB_W, gens[b]);
}

struct DevLayout { MsmLayout L; };

static void run_msm(const MsmLayout& L, uint32_t rows, const uint32_t* digits, std::vector<uint32_t>& partial) {
    MsmView m; m.rows = rows; m.nslots = L.nslots(); m.nchunks = L.nchunks(); m.table = g_table.data(); m.digits = digits;
    m.slot_base = L.slot_base.data(); m.slot_scalar = nullptr; m.slot_nwin = L.slot_nwin.data(); m.chunk_begin = L.chunk_begin.data(); m.chunk_win0 = L.chunk_win0.data(); m.chunk_nwin = L.chunk_nwin.data();
    partial.assign((size_t)L.nchunks() * GE_W * rows, 0); m.partial = partial.data(); m.acc_init = nullptr;
    for (uint32_t c = 0; c < L.nchunks(); c++) for (uint32_t row = 0; row < rows; row++) msm_chunk_ref(m, c, row);
}
static void run_reduce(const MsmLayout& L, uint32_t rows, const std::vector<uint32_t>& partial, uint32_t* enc, const uint64_t* out_off, uint8_t* out) {
    ReduceView r; r.rows = rows; r.ntargets = L.ntargets(); r.partial = partial.data(); r.target_chunk_begin = L.target_chunk_begin.data();
    r.enc = enc; r.out_off = out_off; r.out = out; r.corr = nullptr;
    for (uint32_t t = 0; t < L.ntargets(); t++) for (uint32_t row = 0; row < rows; row++) reduce_encode_thread(r, t, row);
}

// (e + 1) * 2^(16 w) * G by plain double-and-add, as packed affine-Niels words: the independent route the table builder is checked against
static void ref_entry(const ge& g, uint32_t w, uint32_t e, uint32_t out[EDG_ENTRY_W]) {
    ge p = g;
    for (uint32_t k = 0; k < EDG_WBITS * w; k++) p = ge_dbl(p);
    ge acc = ge_identity();
    for (int bit = 15; bit >= 0; bit--) { acc = ge_dbl(acc); if (((e + 1) >> bit) & 1u) acc = ge_add(acc, p); }
    const fe zi = host_fe_invert(acc.Z);
    const fe x = fe_mul(acc.X, zi), y = fe_mul(acc.Y, zi);
    fe_towords(out, fe_add(y, x)); fe_towords(out + 8, fe_sub(y, x)); fe_towords(out + 16, fe_mul(fe_mul(x, y), fe_const_d2()));
}

extern "C" {
void emul_sc_recode65536(const uint32_t raw[8], uint32_t out[8]) { sc r; for (int k = 0; k < 8; k++) r.v[k] = raw[k]; sc_recode_signed65536(out, r); }

// Builds window w (all 32 768 slots) of generator `gen` with the device builder's own steps (edg.h: bases, run starts, fill, batched
// conversion), plus the first slots of window w + 1, runs the builder's self-check over the window, and compares `nsample` slots
// (first, last, seeded others) with double-and-add.  Returns the number of mismatches of either kind.
int emul_edg_window(uint32_t gen, uint32_t w, uint32_t nsample, uint32_t seed) {
    ge gens[NBASE]; host_generators(gens);
    std::vector<uint32_t> g1(GE_W); st_ge(g1.data(), 0, 0, 1, gens[gen]);
    std::vector<uint32_t> bases((size_t)EDG_NWIN * GE_W), starts((size_t)EDG_NWIN * EDG_NSEG * GE_W), table((size_t)EDG_NWIN * EDG_NENT * EDG_SLOT_W, 0xDEADBEEFu);
    edg_step_bases(g1.data(), bases.data(), 0);
    const uint32_t last = w + 1 < EDG_NWIN ? w + 1 : w;
    for (uint32_t ww = w; ww <= last; ww++) {
        edg_step_starts(bases.data(), starts.data(), ww);
        const uint32_t runs = ww == w ? EDG_NSEG : 1;
        for (uint32_t s = 0; s < runs; s++) edg_step_fill(bases.data(), starts.data(), table.data(), ww, s);
        for (size_t g = 0; g < (size_t)runs * EDG_SEG / EDG_INV; g++) edg_step_affine(table.data(), (size_t)ww * EDG_NENT / EDG_INV + g);
    }
    int bad = 0;
    for (uint32_t e = 0; e < EDG_NENT; e++) if (!edg_step_check(table.data(), g1.data(), 0, w, e)) bad++;
    uint32_t x = seed * 2654435761u + 12345u;
    for (uint32_t k = 0; k < nsample; k++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t e = k == 0 ? 0 : k == 1 ? EDG_NENT - 1 : (x >> 8) % EDG_NENT;
        uint32_t want[EDG_ENTRY_W]; ref_entry(gens[gen], w, e, want);
        if (memcmp(want, table.data() + edg_slot(0, w, e), sizeof want) != 0) bad++;
        for (uint32_t pad = EDG_ENTRY_W; pad < EDG_SLOT_W; pad++) if (table[edg_slot(0, w, e) + pad] != 0) bad++;
    }
    // the self-check must also FIND a broken slot
    table[edg_slot(0, w, 77) + 3] ^= 4u;
    if (edg_step_check(table.data(), g1.data(), 0, w, 76) && edg_step_check(table.data(), g1.data(), 0, w, 77)) bad++;
    return bad;
}
// one fixed-base term through the gather path's own pieces: sum_w digit_w * 2^(16 w) * G from radix-2^16 digits and entries built on the
// spot, against k * G by double-and-add (ristretto encodings compared).  Returns 1 when equal.
int emul_edg_term(uint32_t gen, const uint32_t raw[8], uint32_t enc_out[8]) {
    ge gens[NBASE]; host_generators(gens);
    sc r; for (int k = 0; k < 8; k++) r.v[k] = raw[k];
    uint32_t dig[8]; sc_recode_signed65536(dig, r);
    ge acc = ge_identity();
    for (uint32_t w = 0; w < EDG_NWIN; w++) {
        const int32_t d = (int32_t)(int16_t)(dig[w >> 1] >> (16 * (w & 1u)));
        if (d == 0) continue;
        uint32_t entry[EDG_ENTRY_W]; ref_entry(gens[gen], w, (uint32_t)(d < 0 ? -d : d) - 1, entry);
        acc = edg_accumulate(acc, d, entry);
    }
    ge want = ge_identity();
    for (int bit = 255; bit >= 0; bit--) { want = ge_dbl(want); if ((raw[bit >> 5] >> (bit & 31)) & 1u) want = ge_add(want, gens[gen]); }
    uint32_t a[8], b[8]; ge_ristretto_encode(a, acc); ge_ristretto_encode(b, want);
    memcpy(enc_out, a, 32);
    return memcmp(a, b, 32) == 0;
}
void emul_generator(uint32_t idx, uint32_t enc[8]) { ge g[NBASE]; host_generators(g); ge_ristretto_encode(enc, g[idx]); }

// same contract as zkp_hip_prove_range_batch (include/libzkp_hip.h)
int emul_prove_range_batch_bits(uint64_t n, const uint64_t* value, const uint64_t* mn, const uint64_t* mx, uint32_t n_bits, const uint8_t* seeds,
                                uint8_t* out, uint64_t stride, uint32_t* out_len, int32_t* status, uint32_t win_budget) {
    ensure_table();
    uint32_t lg = 0; while ((1u << lg) < n_bits) lg++;
    if (lg < 3 || lg > 6 || (1u << lg) != n_bits) return -2;
    const uint32_t M = (uint32_t)(2 * n), C = (uint32_t)n;
    std::vector<uint64_t> v(M), poff(M), coff(M), ctv(C), ctoff(C);
    std::vector<uint32_t> six(M), pix(M), ctsix(C), ctbl(C);
    std::vector<int32_t> blp(M), blm(M);
    std::vector<uint8_t> kind(M);
    JobBuf J{v.data(), six.data(), pix.data(), blp.data(), blm.data(), kind.data(), poff.data(), coff.data(), ctv.data(), ctsix.data(), ctbl.data(), ctoff.data()};
    for (uint32_t op = 0; op < n; op++) step_build_range(J, op, value, mn, mx, lg, out, stride, out_len, status);
    std::vector<uint32_t> seedw(8 * n); memcpy(seedw.data(), seeds, 32 * n);
    auto words = [&](size_t k) { return std::vector<uint32_t>(k * 8 * M, 0xDEADBEEFu); };
    auto dwords = [&](size_t k) { return std::vector<uint32_t>(k * DIGW * M, 0xDEADBEEFu); };
    auto d1 = dwords(P1_NSLOTS), d2 = dwords(P2_NSLOTS), dr = dwords(PR_NSLOTS);
    auto tape = words(TAPE_SLOTS), gamma = words(1), yinv = words(64), ypq = words(32),
         r0 = words(64), r1 = words(64), pp = words(192), ab = words(256), gh = words(128), scal = words(SC_NUM), enc = words(3);
    std::vector<uint32_t> tstate((size_t)52 * M);
    BpView V; V.M = M; V.n = n_bits; V.lg = lg; V.v = v.data(); V.seed_ix = six.data(); V.proof_ix = pix.data(); V.bl_plus = blp.data(); V.bl_minus = blm.data(); V.kind = kind.data();
    V.seeds = seedw.data(); V.proof_off = poff.data(); V.commit_off = coff.data(); V.out = out;
    V.tape = tape.data(); V.gamma = gamma.data(); V.d1 = d1.data(); V.d2 = d2.data(); V.dr = dr.data(); V.yinvpow = yinv.data(); V.ypq = ypq.data();
    V.r0 = r0.data(); V.r1 = r1.data(); V.pp = pp.data(); V.ab = ab.data(); V.gh = gh.data(); V.scal = scal.data(); V.tstate = tstate.data(); V.enc = enc.data();
    std::vector<uint32_t> partial;
    uint32_t st[50]; Strobe s; s.base = st; s.stride = 1;

    // commitment tasks
    std::vector<uint32_t> ctd((size_t)2 * DIGW * C), ctenc((size_t)8 * C);
    CtView T{C, ctv.data(), ctsix.data(), ctbl.data(), seedw.data(), ctd.data()};
    for (uint32_t c = 0; c < C; c++) step_ctask(T, c);
    const bool even = win_budget >= 10000; const uint32_t nch = win_budget - 10000;
    MsmLayout Lc = even ? make_layout_even(targets_ctask(), nch) : layout_ctask(win_budget);
    run_msm(Lc, C, ctd.data(), partial); run_reduce(Lc, C, partial, ctenc.data(), ctoff.data(), out);

    for (uint32_t slot = 0; slot <= tape_slots(n_bits); slot++) for (uint32_t j = 0; j < M; j++) step_tape(V, slot, j);
    MsmLayout L1 = even ? make_layout_even(targets_phase1(n_bits), nch) : layout_phase1(win_budget, n_bits);
    run_msm(L1, M, V.d1, partial); run_reduce(L1, M, partial, V.enc, nullptr, nullptr);
    for (uint32_t j = 0; j < M; j++) step_transcript1(V, j, s);
    for (uint32_t i = 0; i < n_bits; i++) for (uint32_t j = 0; j < M; j++) step_poly(V, i, j);
    for (uint32_t j = 0; j < M; j++) step_poly_sum(V, j);
    MsmLayout L2 = even ? make_layout_even(targets_phase2(), nch) : layout_phase2(win_budget);
    run_msm(L2, M, V.d2, partial); run_reduce(L2, M, partial, V.enc, nullptr, nullptr);
    for (uint32_t j = 0; j < M; j++) step_transcript2(V, j, s);
    for (uint32_t i = 0; i < n_bits; i++) for (uint32_t j = 0; j < M; j++) step_lr_init(V, i, j);
    for (uint32_t r = 0; r < lg; r++) {
        for (uint32_t i = 0; i < n_bits; i++) for (uint32_t j = 0; j < M; j++) step_round_prep(V, r, i, j);
        for (uint32_t j = 0; j < M; j++) step_round_sum(V, r, j);
        MsmLayout Lr = even ? make_layout_even(targets_round(r, n_bits), nch) : layout_round(r, win_budget, n_bits);
        run_msm(Lr, M, V.dr, partial); run_reduce(Lr, M, partial, V.enc, nullptr, nullptr);
        for (uint32_t j = 0; j < M; j++) step_transcript_round(V, r, j, s);
    }
    int fail = 0; for (uint32_t op = 0; op < n; op++) fail |= status[op] != 0;
    return fail;
}
int emul_prove_range_batch(uint64_t n, const uint64_t* value, const uint64_t* mn, const uint64_t* mx, const uint8_t* seeds,
                           uint8_t* out, uint64_t stride, uint32_t* out_len, int32_t* status, uint32_t win_budget) {
    return emul_prove_range_batch_bits(n, value, mn, mx, 64, seeds, out, stride, out_len, status, win_budget);
}

// same contract as zkp_hip_verify_range_batch (include/libzkp_hip.h); nchunks = chunk count of the fixed-base part
int emul_verify_range_batch(uint64_t n, const uint8_t* proofs, uint64_t stride, const uint32_t* lens, const uint64_t* mins, const uint64_t* maxs,
                            uint8_t* ok, uint32_t nchunks) {
    ensure_table();
    const uint32_t M = (uint32_t)(2 * n);
    const MsmLayout L = make_layout_even(targets_verify(), nchunks);
    std::vector<uint64_t> poff(M), voff(M); std::vector<uint8_t> kind(M), lgn(M); std::vector<int32_t> bad(M);
    std::vector<uint32_t> pts((size_t)VP_NUM * GE_W * M), scal((size_t)VS_NUM * 8 * M), dig((size_t)NBASE * DIGW * M, 0), vs((size_t)VP_NUM * 8 * M, 0);
    std::vector<uint32_t> partial((size_t)(L.nchunks() + VP_NUM) * GE_W * M), enc((size_t)8 * M);
    VfyView V{}; V.M = M; V.in = proofs; V.proof_off = poff.data(); V.venc_off = voff.data(); V.kind = kind.data(); V.lgn = lgn.data(); V.bad = bad.data();
    V.pts = pts.data(); V.scal = scal.data(); V.digits = dig.data(); V.vscal = vs.data(); V.partial = partial.data(); V.var_chunk0 = L.nchunks(); V.table = g_table.data();
    for (uint32_t i = 0; i < n; i++) step_vparse(V, i, proofs + stride * i, stride * i, lens[i] <= stride ? lens[i] : 0u, mins[i], maxs[i]);
    for (uint32_t p = 0; p < VP_NUM; p++) for (uint32_t j = 0; j < M; j++) step_vdecode(V, p, j);
    uint32_t st[50]; Strobe s; s.base = st; s.stride = 1;
    for (uint32_t j = 0; j < M; j++) { s.pos = 0; s.pos_begin = 0; step_vtranscript(V, j, s); }
    for (uint32_t i = 0; i < BP_N; i++) for (uint32_t j = 0; j < M; j++) step_vscalars(V, i, j);
    MsmView m; m.rows = M; m.nslots = L.nslots(); m.nchunks = L.nchunks(); m.table = g_table.data(); m.digits = dig.data();
    m.slot_base = L.slot_base.data(); m.slot_scalar = nullptr; m.slot_nwin = L.slot_nwin.data(); m.chunk_begin = L.chunk_begin.data(); m.chunk_win0 = L.chunk_win0.data(); m.chunk_nwin = L.chunk_nwin.data();
    m.partial = partial.data(); m.acc_init = nullptr;
    for (uint32_t c = 0; c < L.nchunks(); c++) for (uint32_t j = 0; j < M; j++) msm_chunk_ref(m, c, j);
    for (uint32_t p = 0; p < VP_NUM; p++) for (uint32_t j = 0; j < M; j++) step_vvarbase(V, p, j);
    const uint16_t tcb[2] = {0, (uint16_t)(L.nchunks() + VP_NUM)};
    ReduceView r; r.rows = M; r.ntargets = 1; r.partial = partial.data(); r.target_chunk_begin = tcb; r.enc = enc.data(); r.out_off = nullptr; r.out = nullptr; r.corr = nullptr;
    for (uint32_t j = 0; j < M; j++) reduce_encode_thread(r, 0, j);
    for (uint32_t i = 0; i < n; i++) step_vfinal(V, enc.data(), i, ok, 2);
    return 0;
}
// k * P on ristretto encodings (decode, signed radix-4 multiplication, encode); 0 if the encoding is invalid
int emul_scalarmult(const uint32_t enc_in[8], const uint32_t k[8], uint32_t enc_out[8]) {
    ge p; if (!ge_ristretto_decode(p, enc_in)) return 0;
    sc kk; memcpy(kk.v, k, 32);
    ge_ristretto_encode(enc_out, ge_scalarmult_raw(p, kk));
    return 1;
}
}
