// TEST INFRASTRUCTURE: host run of the Groth16 witness + QAP steps (the per-proof work before the MSMs) so the no-GPU
// tier can compare the full assignment z and the quotient coefficients h with the oracle.
#include "../../libzkp_amd/csrc/g16_circuit.h"
#include "../../libzkp_amd/csrc/g16_rlc.h"
#include "../../libzkp_amd/csrc/fq2vm.h"
#include <vector>
using namespace zkp;

struct NoSync { void operator()() const {} };

extern "C" {
// kind 0 equality / 1 membership.  Outputs: z_raw [nv][8] canonical words, h_raw [m-1][8], envelope prefix bytes.
// h is returned through its signed digits (what the MSM consumes), re-assembled to the canonical value.
int emul_g16_witness_qap(int kind, uint64_t value, const uint64_t* set_vals, uint32_t set_len, const uint8_t seed[32],
                         uint32_t* z_raw, uint32_t* h_raw, uint32_t* rs_raw, uint8_t* out, uint64_t stride, uint32_t* shape, uint32_t radix_bits) {
    const HostR1CS cs = kind == 0 ? build_equality_r1cs() : build_membership_r1cs();
    const HostCircuitTables T = build_circuit_tables(cs);
    ensure_mimc_constants();
    std::vector<uint32_t> mc; for (auto& c : g_mimc_host) put_fr(mc, c);
    const uint32_t nsc = g16_nscalars(T.nv, T.m);
    const G16Radix rx = radix_bits == 114 ? g16_radix(14, true) : g16_radix(radix_bits ? radix_bits : G16_WBITS_DEFAULT);          // 114: the uneven form of radix 2^14
    std::vector<uint32_t> z((size_t)T.nv * 8), sdig((size_t)nsc * rx.digw), rs(16), seedw(8);
    memcpy(seedw.data(), seed, 32);
    uint64_t sv[G16_MAX_SET] = {0}; for (uint32_t i = 0; i < set_len && i < G16_MAX_SET; i++) sv[i] = set_vals[i];
    G16View V{}; V.rx = rx; V.rows = 1; V.kind = (uint32_t)kind; V.n_inst = T.n_inst; V.n_wit = T.n_wit; V.nv = T.nv; V.m = T.m;
    V.value = &value; V.set_vals = sv; V.set_len = &set_len; V.seeds = seedw.data(); V.mimc_c = mc.data();
    V.z = z.data(); V.sdig = sdig.data(); V.rs = rs.data(); V.out = out; V.stride = stride;
    step_g16_witness(V, 0);
    for (uint32_t k = 0; k < T.nv; k++) step_g16_zdigits(V, k, 0);
    G16Circuit C{}; C.n_rows = T.n_rows; C.m = T.m; C.logm = T.logm;
    C.a_ptr = T.ptr[0].data(); C.a_col = T.col[0].data(); C.a_coef = T.coef[0].data();
    C.b_ptr = T.ptr[1].data(); C.b_col = T.col[1].data(); C.b_coef = T.coef[1].data();
    C.c_ptr = T.ptr[2].data(); C.c_col = T.col[2].data(); C.c_coef = T.coef[2].data();
    C.tw = T.tw.data(); C.tw_inv = T.tw_inv.data(); C.coset = T.coset.data(); C.coset_inv = T.coset_inv.data(); C.zinv = T.zinv.data();
    std::vector<uint32_t> lds((size_t)9 * T.m), scratch((size_t)2 * 9 * T.m);
    G16Lds L; L.base = lds.data(); L.m = T.m;
    g16_qap_proof(V, C, L, scratch.data(), 0, 0, 1, NoSync());
    for (uint32_t k = 0; k < T.nv; k++) { fr x = ld_fr(z.data(), k, 0, 1); fp_to_raw(z_raw + 8 * k, x); }
    // digits -> canonical value (sum d_j 2^(WBITS j)), as 8 words
    auto undigit = [&](uint32_t idx, uint32_t* outw) {
        unsigned __int128 lo = 0, hi = 0;     // 260-bit accumulator as two halves: value = hi * 2^128 + lo (two's complement overall)
        // Horner from the top digit: acc = acc * 2^(distance to the next window) + d
        for (int j = (int)rx.nwin - 1; j >= 0; j--) {
            const int32_t d = (int32_t)(int16_t)(sdig[(size_t)idx * rx.digw + (j >> 1)] >> (16 * (j & 1)));
            const uint32_t sh = j + 1 < (int)rx.nwin ? g16_win_bit(rx, (uint32_t)j + 1) - g16_win_bit(rx, (uint32_t)j) : rx.wbits;
            hi = (hi << sh) | (lo >> (128 - sh)); lo <<= sh;
            if (d >= 0) { const unsigned __int128 t = lo + (unsigned)d; if (t < lo) hi++; lo = t; }
            else { const unsigned __int128 t = lo - (unsigned)(-d); if (t > lo) hi--; lo = t; }
        }
        for (int k = 0; k < 4; k++) { outw[k] = (uint32_t)(lo >> (32 * k)); outw[4 + k] = (uint32_t)(hi >> (32 * k)); }
    };
    for (uint32_t i = 0; i + 1 < T.m; i++) undigit(g16_sc_h(V) + i, h_raw + 8 * i);
    memcpy(rs_raw, rs.data(), 64);
    shape[0] = T.n_inst; shape[1] = T.n_wit; shape[2] = T.n_rows; shape[3] = T.m;
    return 0;
}

// Groth16 verification of one envelope (kind 0 equality / 1 membership) under a verifying key given as raw affine
// coordinates: alpha (16 words), beta / gamma / delta (32 each), ic (n_ic x 16)
// window tables of one gamma_abc point, as k_g16_build_table builds them (the device builder's batched conversion): windows 0 .. nwin - 1
static void build_ic_windows(uint32_t* tab, uint32_t i, const g1_aff& p, uint32_t nwin) {
    g1_jac q = jac_from_aff(p);
    for (uint32_t w = 0; w < nwin; w++) {
        g1_jac acc = q;
        for (uint32_t e0 = 0; e0 < G16V_NENT; e0 += 8) {
            g1_jac pts[8]; fq zp[8];
            for (uint32_t k = 0; k < 8; k++) { pts[k] = acc; acc = jac_add(acc, q); zp[k] = k ? f_mul(zp[k - 1], pts[k].Z) : pts[k].Z; }
            fq inv = f_inv(zp[7]);
            for (int k = 7; k >= 0; k--) {
                const fq zi = k ? f_mul(inv, zp[k - 1]) : inv;
                if (k) inv = f_mul(inv, pts[k].Z);
                const fq zi2 = f_sq(zi);
                g1_aff a; a.x = f_mul(pts[k].X, zi2); a.y = f_mul(pts[k].Y, f_mul(zi2, zi));
                uint32_t* dst = tab + (((size_t)i * G16V_NWIN + w) * G16V_NENT + e0 + k) * 20;
                for (int j = 0; j < 10; j++) { dst[j] = a.x.v[j]; dst[10 + j] = a.y.v[j]; }
            }
        }
        for (uint32_t k = 0; k < G16V_WBITS; k++) q = jac_dbl(q);
    }
}
// mode 0: g16_verify.h's lane-per-chain code; mode 4: the Fq2 machine (fq2vm.h) with its tables (four waves per chain),
// every round's streams executed as the device executes them.  Returns the verdict, or 2 when the machine leaves the envelope to the
// lane-per-chain path (a point at infinity in the proof).
static int verify_with(int mode, int kind, const uint8_t* env, uint32_t len, const uint32_t* alpha, const uint32_t* beta, const uint32_t* gamma, const uint32_t* delta,
                       uint32_t n_ic, const uint32_t* ic) {
    auto g1 = [](const uint32_t* w) { return g1_aff{fq_from_raw(w), fq_from_raw(w + 8)}; };
    auto g2 = [](const uint32_t* w) { return g2_aff{fq2{fq_from_raw(w), fq_from_raw(w + 8)}, fq2{fq_from_raw(w + 16), fq_from_raw(w + 24)}}; };
    std::vector<uint32_t> icm((size_t)n_ic * 20);
    for (uint32_t i = 0; i < n_ic; i++) { const g1_aff p = g1(ic + 16 * i); for (int k = 0; k < 10; k++) { icm[20 * i + k] = p.x.v[k]; icm[20 * i + 10 + k] = p.y.v[k]; } }
    G16Vk vk; vk.gamma = g2(gamma); vk.delta = g2(delta); vk.n_ic = n_ic; vk.ic = icm.data(); vk.ic_table = nullptr;
    // small keys (equality: two points) also get the window tables the GPU path uses, built as k_g16_build_table builds them
    std::vector<uint32_t> tab;
    if (n_ic <= 2) {
        tab.resize((size_t)n_ic * G16V_NWIN * G16V_NENT * 20);
        for (uint32_t i = 0; i < n_ic; i++) build_ic_windows(tab.data(), i, g1(ic + 16 * i), G16V_NWIN);
        vk.ic_table = tab.data();
    }
    if (mode == 0) {
        vk.ml_alpha_beta = miller_loop(g2(beta), aff_neg(g1(alpha)));
        return g16_verify_envelope(kind, vk, env, len) ? 1 : 0;
    }
    namespace vm = fq2vm;
    G16Pairs o;
    g1_jac Lj;                                           // as k_g16_pairs_vm: the public-input point stays Jacobian
    if (!(kind == G16_EQUALITY ? g16_equality_pairs(vk, env, len, o, &Lj) : g16_membership_pairs(vk, env, len, o, &Lj))) return 0;
    if (o.present != 15u) return 2;
    const vm::Tables T = vm::Tables{vm::CODE_K4, vm::OFF_K4, 4, &vm::CONSTS[0][0]};
    const uint32_t* regs = vm::REGS_K4;
    std::vector<uint32_t> io((size_t)vm::N_SLOTS * vm::FQ2_W, 0u), kc(6 * vm::FQ2_W);
    auto put = [&](std::vector<uint32_t>& buf, uint32_t slot, const fq2& x) { vm::fq2_to_words(&buf[(size_t)slot * vm::FQ2_W], x); };
    auto get = [&](uint32_t slot) { return vm::fq2_from_words(&io[(size_t)slot * vm::FQ2_W]); };
    std::vector<uint32_t> lines;           // what the library computes when a key is loaded (fq2vm_kernels.hip does the same through g16_vm_key_constants)
    {
        auto run1 = [&](const uint16_t* script, uint32_t len, uint32_t regs_ix, std::vector<uint32_t>& buf) { vm::Launch L{T, script, len, 1, buf.data(), 0, nullptr}; vm::run_host(L, 0, regs[regs_ix]); };
        std::vector<uint32_t> io2((size_t)vm::N_SLOTS * vm::FQ2_W, 0u);
        const g2_aff b = g2(beta); const g1_aff na = aff_neg(g1(alpha));
        put(io2, vm::SLOT_QX, b.x); put(io2, vm::SLOT_QY, b.y); put(io2, vm::SLOT_P, fq2{na.x, na.y});
        run1(vm::SCRIPT_MILLER, sizeof(vm::SCRIPT_MILLER) / 2, 0, io2);
        for (uint32_t k = 0; k < 6 * vm::FQ2_W; k++) kc[k] = io2[(size_t)vm::SLOT_F0 * vm::FQ2_W + k];
        lines.assign((size_t)vm::N_LINE_SLOTS * vm::FQ2_W, 0u);
        for (int j = 0; j < 2; j++) {
            std::vector<uint32_t> io3((size_t)(vm::LINE_SLOT0 + vm::N_LINE_SLOTS) * vm::FQ2_W, 0u);
            put(io3, vm::SLOT_QX, j ? vk.delta.x : vk.gamma.x); put(io3, vm::SLOT_QY, j ? vk.delta.y : vk.gamma.y);
            run1(vm::SCRIPT_LINES, sizeof(vm::SCRIPT_LINES) / 2, 4, io3);
            for (uint32_t st = 0; st < vm::N_LINE_SLOTS / 6; st++)
                for (uint32_t k = 0; k < 3 * vm::FQ2_W; k++) lines[((size_t)6 * st + 3 * j) * vm::FQ2_W + k] = io3[((size_t)vm::LINE_SLOT0 + 6 * st) * vm::FQ2_W + k];
        }
    }
    for (uint32_t j = 0; j < 3; j++) {
        if (j == 1) { fq2 p1, p1z; g16_vm_pair1(Lj, p1, p1z); put(io, vm::PAIR_SLOTS + vm::SLOT_P, p1); put(io, vm::PAIR_SLOTS + vm::SLOT_QX, p1z); continue; }
        put(io, vm::PAIR_SLOTS * j + vm::SLOT_QX, o.Q[j].x); put(io, vm::PAIR_SLOTS * j + vm::SLOT_QY, o.Q[j].y); put(io, vm::PAIR_SLOTS * j + vm::SLOT_P, fq2{o.P[j].x, o.P[j].y});
    }
    { vm::Launch L{T, vm::SCRIPT_MILLER, (uint32_t)(sizeof(vm::SCRIPT_MILLER) / 2), 1, io.data(), 0, nullptr}; vm::run_host(L, 0, regs[0]); }
    { vm::Launch L{T, vm::SCRIPT_MILLER_B, (uint32_t)(sizeof(vm::SCRIPT_MILLER_B) / 2), 1, io.data() + (size_t)vm::PAIR_SLOTS * vm::FQ2_W, 0, lines.data()}; vm::run_host(L, 0, regs[3]); }
    { vm::Launch L{T, vm::SCRIPT_SUBGROUP, (uint32_t)(sizeof(vm::SCRIPT_SUBGROUP) / 2), 1, io.data(), 0, nullptr}; vm::run_host(L, 0, regs[1]); }
    { vm::Launch L{T, vm::SCRIPT_FINISH, (uint32_t)(sizeof(vm::SCRIPT_FINISH) / 2), 1, io.data(), 0, kc.data()}; vm::run_host(L, 0, regs[2]); }
    bool good = !f_is_zero(get(vm::SLOT_SZ)) && f_is_zero(get(vm::SLOT_SH)) && f_is_zero(get(vm::SLOT_SR));
    good = good && fq2_eq(get(vm::SLOT_RES), fq2_one());
    for (uint32_t k = 1; k < 6; k++) good = good && f_is_zero(get(vm::SLOT_RES + k));
    return good ? 1 : 0;
}
int emul_g16_verify(int kind, const uint8_t* env, uint32_t len, const uint32_t* alpha, const uint32_t* beta, const uint32_t* gamma, const uint32_t* delta,
                    uint32_t n_ic, const uint32_t* ic) { return verify_with(0, kind, env, len, alpha, beta, gamma, delta, n_ic, ic); }
int emul_g16_verify_vm(int waves, int kind, const uint8_t* env, uint32_t len, const uint32_t* alpha, const uint32_t* beta, const uint32_t* gamma, const uint32_t* delta,
                       uint32_t n_ic, const uint32_t* ic) { return verify_with(4, kind, env, len, alpha, beta, gamma, delta, n_ic, ic); }
// The public-input point of an envelope on `nlanes` cooperating lanes (g16_public_input_lane: what k_g16_public_inputs runs, window tables
// built as the device builds them) against the one-lane form without tables.  1: the lanes' partial points sum to the same point; 0: they
// do not; -1: the header is refused.
int emul_g16_public_input_lanes(int kind, const uint8_t* env, uint32_t len, uint32_t n_ic, const uint32_t* ic, uint32_t nlanes) {
    auto g1 = [](const uint32_t* w) { return g1_aff{fq_from_raw(w), fq_from_raw(w + 8)}; };
    std::vector<uint32_t> icm((size_t)n_ic * 20);
    for (uint32_t i = 0; i < n_ic; i++) { const g1_aff p = g1(ic + 16 * i); for (int k = 0; k < 10; k++) { icm[20 * i + k] = p.x.v[k]; icm[20 * i + 10 + k] = p.y.v[k]; } }
    G16Vk vk; vk.n_ic = n_ic; vk.ic = icm.data(); vk.ic_table = nullptr;
    G16Inputs h;
    if (!g16_header(kind, vk, env, len, h)) return -1;
    const g1_jac want = g16_public_input_point(vk, h);
    std::vector<uint32_t> tab((size_t)n_ic * G16V_NWIN * G16V_NENT * 20, 0u);          // only the windows the envelope's scalars reach are filled
    build_ic_windows(tab.data(), 1, g1(ic + 16), G16V_NWIN);
    for (uint32_t i = 0; i < h.n; i++) {
        const uint64_t v = g16_set_element(h, i);
        uint32_t bits = 0; while (bits < 64 && (v >> bits) != 0) bits++;
        const uint32_t nw = bits / G16V_WBITS + 2;          // the windows a value of that size can reach (the signed digits carry one window up)
        build_ic_windows(tab.data(), 2 + i, g1(ic + 16 * (2 + i)), nw < G16V_NWIN_U64 ? nw : G16V_NWIN_U64);
    }
    vk.ic_table = tab.data();
    g1_jac sum = jac_infinity<fq>();
    for (uint32_t lane = nlanes; lane-- > 0;) sum = jac_add(g16_public_input_lane(vk, h, lane, nlanes), sum);
    g1_aff a, b;
    const bool fa = jac_to_aff(a, want), fb = jac_to_aff(b, sum);
    if (fa != fb) return 0;
    return !fa || (fq_eq(a.x, b.x) && fq_eq(a.y, b.y)) ? 1 : 0;
}
// g16_rlc.h: one pairing check for `count` envelopes (rows of `envs`, `stride` bytes apart) under the weights rho (4 words each), on the
// lane-per-chain pairing code.  1 accepted, 0 rejected, 2 left to the per-envelope check.
int emul_g16_rlc(int kind, uint32_t count, const uint8_t* envs, uint32_t stride, const uint32_t* lens, const uint32_t* rho, const uint32_t* alpha, const uint32_t* beta,
                 const uint32_t* gamma, const uint32_t* delta, uint32_t n_ic, const uint32_t* ic) {
    auto g1 = [](const uint32_t* w) { return g1_aff{fq_from_raw(w), fq_from_raw(w + 8)}; };
    auto g2 = [](const uint32_t* w) { return g2_aff{fq2{fq_from_raw(w), fq_from_raw(w + 8)}, fq2{fq_from_raw(w + 16), fq_from_raw(w + 24)}}; };
    std::vector<uint32_t> icm((size_t)n_ic * 20);
    for (uint32_t i = 0; i < n_ic; i++) { const g1_aff p = g1(ic + 16 * i); for (int k = 0; k < 10; k++) { icm[20 * i + k] = p.x.v[k]; icm[20 * i + 10 + k] = p.y.v[k]; } }
    G16Vk vk; vk.gamma = g2(gamma); vk.delta = g2(delta); vk.beta = g2(beta); vk.n_ic = n_ic; vk.ic = icm.data(); vk.ic_table = nullptr;
    vk.ml_alpha_beta = miller_loop(g2(beta), aff_neg(g1(alpha)));
    std::vector<const uint8_t*> ptr(count);
    for (uint32_t j = 0; j < count; j++) ptr[j] = envs + (size_t)stride * j;
    return g16_rlc_check(kind, vk, g1(alpha), g2(beta), count, ptr.data(), lens, rho);
}
}
