// Host-side description of libzkp's two Groth16 circuits as R1CS (shape only: which variable appears in which row with
// which coefficient).  Mirrors EqualityCircuit::generate_constraints (/root/reference/src/backend/snark.rs:262-291),
// MembershipCircuit::generate_constraints (:514-585) and mimc_hash_circuit (:232-247) including the allocation order of
// instance and witness variables.  One-time setup data; witness VALUES are produced on the GPU (g16_steps.h).
#pragma once
#include <mutex>
#include <array>
#include <map>
#include <vector>
#include <cstring>
#include "g16_steps.h"
#include "sha256_host.h"

namespace zkp {

// ---- host Fr helpers (one-time setup math; the same host+device field code)
inline fr fr_pow_words(fr base, const uint32_t e[8]) {
    fr acc = fp_one<FrParams>();
    for (int i = 255; i >= 0; i--) { acc = fp_sq(acc); if ((e[i >> 5] >> (i & 31)) & 1u) acc = fp_mul(acc, base); }
    return acc;
}
inline void put_fr(std::vector<uint32_t>& v, const fr& x) { for (int k = 0; k < 8; k++) v.push_back(x.v[k]); }

// ---- R1CS description of the two circuits (shape only; values are produced on the GPU by step_g16_witness)
struct HostLC { std::map<uint32_t, fr> t; };       // column -> coefficient
struct HostR1CS {
    uint32_t n_inst = 1, n_wit = 0;
    std::vector<std::array<HostLC, 3>> rows;
    std::vector<uint8_t> inst_nwin{1}, wit_nwin;   // windows a variable's scalar can occupy: a literal count, or one of the two classes below
    uint32_t new_input(uint8_t nwin) { inst_nwin.push_back(nwin); return n_inst++; }
    uint32_t new_witness(uint8_t nwin) { wit_nwin.push_back(nwin); return 0x80000000u | n_wit++; }
};
constexpr uint32_t WIT = 0x80000000u;
// size classes of a variable's scalar; the window count follows from the radix of the key's tables when the key is loaded
constexpr uint8_t G16_NW_FULL = 0xFF, G16_NW_U64 = 0xFE;
inline uint8_t g16_class_nwin(uint8_t cls, const G16Radix& rx) { return cls == G16_NW_FULL ? (uint8_t)rx.nwin : cls == G16_NW_U64 ? (uint8_t)rx.nwin_u64 : cls; }
inline HostLC lc_var(uint32_t v) { HostLC l; l.t[v] = fp_one<FrParams>(); return l; }
inline HostLC lc_add(const HostLC& a, const HostLC& b) {
    HostLC o = a;
    for (auto& kv : b.t) { auto it = o.t.find(kv.first); if (it == o.t.end()) o.t[kv.first] = kv.second; else it->second = fp_add(it->second, kv.second); }
    for (auto it = o.t.begin(); it != o.t.end();) { if (fp_is_zero(it->second)) it = o.t.erase(it); else ++it; }
    return o;
}
inline HostLC lc_scale(const HostLC& a, const fr& s) { HostLC o; for (auto& kv : a.t) { fr c = fp_mul(kv.second, s); if (!fp_is_zero(c)) o.t[kv.first] = c; } return o; }
inline HostLC lc_sub(const HostLC& a, const HostLC& b) { return lc_add(a, lc_scale(b, fp_neg(fp_one<FrParams>()))); }
const uint32_t VAR_ONE = 0;
inline uint32_t r1cs_mul(HostR1CS& cs, const HostLC& a, const HostLC& b, uint8_t nwin) {   // AllocatedFp::mul
    const uint32_t p = cs.new_witness(nwin);
    cs.rows.push_back({a, b, lc_var(p)});
    return p;
}
inline void r1cs_enforce_equal(HostR1CS& cs, const HostLC& a, const HostLC& b) { cs.rows.push_back({lc_sub(a, b), lc_var(VAR_ONE), HostLC{}}); }

inline std::vector<fr> g_mimc_host;     // round constants, snark.rs:186-199
inline std::once_flag g_mimc_once;
inline void ensure_mimc_constants() {       // thread-safe: shards load their keys from parallel host threads
    std::call_once(g_mimc_once, []() {
        for (uint32_t i = 0; i < MIMC_ROUNDS; i++) {
            uint8_t in[23]; memcpy(in, "libzkp_mimc_v1:", 15); for (int k = 0; k < 8; k++) in[15 + k] = (uint8_t)((uint64_t)i >> (8 * k));
            uint8_t h[32]; sha256_host(h, in, 23);
            uint32_t w[8]; memcpy(w, h, 32);
            g_mimc_host.push_back(fp_from_raw<FrParams>(w));     // from_le_bytes_mod_order
        }
    });
}
inline HostLC r1cs_mimc(HostR1CS& cs, HostLC x) {                   // snark.rs:232-247
    ensure_mimc_constants();
    for (uint32_t i = 0; i < MIMC_ROUNDS; i++) {
        const HostLC t = lc_add(x, lc_scale(lc_var(VAR_ONE), g_mimc_host[i]));
        const uint32_t t2 = r1cs_mul(cs, t, t, G16_NW_FULL);
        const uint32_t t4 = r1cs_mul(cs, lc_var(t2), lc_var(t2), G16_NW_FULL);
        x = lc_var(r1cs_mul(cs, lc_var(t4), t, G16_NW_FULL));
    }
    return x;
}
inline HostR1CS build_equality_r1cs() {                              // snark.rs:262-291
    HostR1CS cs;
    const uint32_t a = cs.new_witness(G16_NW_U64), b = cs.new_witness(G16_NW_U64);
    r1cs_enforce_equal(cs, lc_var(a), lc_var(b));
    const HostLC h = r1cs_mimc(cs, lc_var(a));
    const uint32_t c = cs.new_input(G16_NW_FULL);
    r1cs_enforce_equal(cs, h, lc_var(c));
    return cs;
}
inline HostR1CS build_membership_r1cs() {                            // snark.rs:514-585
    HostR1CS cs;
    const uint32_t v = cs.new_witness(G16_NW_U64);
    const HostLC h = r1cs_mimc(cs, lc_var(v));
    const uint32_t c = cs.new_input(G16_NW_FULL);
    r1cs_enforce_equal(cs, h, lc_var(c));
    std::vector<uint32_t> setv, real, sel;
    for (uint32_t i = 0; i < G16_MAX_SET; i++) setv.push_back(cs.new_input(G16_NW_U64));
    for (uint32_t i = 0; i < G16_MAX_SET; i++) { const uint32_t b = cs.new_input(1); cs.rows.push_back({lc_sub(lc_var(VAR_ONE), lc_var(b)), lc_var(b), HostLC{}}); real.push_back(b); }
    for (uint32_t i = 0; i < G16_MAX_SET; i++) { const uint32_t b = cs.new_witness(1); cs.rows.push_back({lc_sub(lc_var(VAR_ONE), lc_var(b)), lc_var(b), HostLC{}}); sel.push_back(b); }
    HostLC total;
    for (uint32_t i = 0; i < G16_MAX_SET; i++) {
        total = lc_add(total, lc_var(sel[i]));
        const uint32_t p = r1cs_mul(cs, lc_var(sel[i]), lc_sub(lc_var(VAR_ONE), lc_var(real[i])), 1);
        r1cs_enforce_equal(cs, lc_var(p), HostLC{});
    }
    r1cs_enforce_equal(cs, total, lc_var(VAR_ONE));
    HostLC acc;
    for (uint32_t i = 0; i < G16_MAX_SET; i++) acc = lc_add(acc, lc_var(r1cs_mul(cs, lc_var(sel[i]), lc_sub(lc_var(v), lc_var(setv[i])), G16_NW_FULL)));
    r1cs_enforce_equal(cs, acc, HostLC{});
    return cs;
}


// CSR form of one matrix (columns: instance block first, then witnesses) + the domain tables of the circuit
struct HostCircuitTables {
    uint32_t n_inst, n_wit, nv, n_rows, m, logm;
    std::vector<uint32_t> ptr[3], col[3], coef[3];
    std::vector<uint32_t> tw, tw_inv, coset, coset_inv, zinv;
};
inline HostCircuitTables build_circuit_tables(const HostR1CS& cs) {
    HostCircuitTables T;
    T.n_inst = cs.n_inst; T.n_wit = cs.n_wit; T.nv = cs.n_inst + cs.n_wit; T.n_rows = (uint32_t)cs.rows.size();
    uint32_t m = 1, lg = 0; while (m < T.n_rows + T.n_inst) { m <<= 1; lg++; }
    T.m = m; T.logm = lg;
    auto colidx = [&](uint32_t v) { return (v & WIT) ? cs.n_inst + (v & ~WIT) : v; };
    for (int q = 0; q < 3; q++) {
        T.ptr[q].push_back(0);
        for (auto& row : cs.rows) {
            for (auto& kv : row[q].t) { T.col[q].push_back(colidx(kv.first)); put_fr(T.coef[q], kv.second); }
            T.ptr[q].push_back((uint32_t)T.col[q].size());
        }
    }
    // domain: w = 5^((r-1)/m), coset offset g = 5 (ark-ff Fr::GENERATOR)
    uint32_t e[8]; for (int i = 0; i < 8; i++) e[i] = FrParams::mod(i);
    e[0] -= 1;
    for (uint32_t s = 0; s < lg; s++) { for (int i = 0; i < 7; i++) e[i] = (e[i] >> 1) | (e[i + 1] << 31); e[7] >>= 1; }
    const fr g = fp_from_u64<FrParams>(5), w = fr_pow_words(g, e);
    const fr winv = fp_inv(w), ginv = fp_inv(g), minv = fp_inv(fp_from_u64<FrParams>(m));
    // the domain tables are read by the QAP step only, which works on nine 29-bit limbs (bn254_fr9.h): nine words per element
    auto put_fr9 = [](std::vector<uint32_t>& v, const fr& x) { const fr9 y = fr9_from_fr(x); for (int i = 0; i < 9; i++) v.push_back(y.v[i]); };
    fr a = fp_one<FrParams>(), b = a;
    for (uint32_t j = 0; j < m / 2; j++) { put_fr9(T.tw, a); put_fr9(T.tw_inv, b); a = fp_mul(a, w); b = fp_mul(b, winv); }
    a = minv; b = minv;
    for (uint32_t i = 0; i < m; i++) { put_fr9(T.coset, a); put_fr9(T.coset_inv, b); a = fp_mul(a, g); b = fp_mul(b, ginv); }
    fr gm = fp_one<FrParams>(); for (uint32_t i = 0; i < m; i++) gm = fp_mul(gm, g);
    put_fr9(T.zinv, fp_inv(fp_sub(gm, fp_one<FrParams>())));
    return T;
}

}  // namespace zkp
