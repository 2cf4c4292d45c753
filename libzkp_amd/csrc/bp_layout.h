// Host-side setup shared by the HIP library and the test emulator: generator derivation
// (PedersenGens::default / BulletproofGens::new party 0, /root/reference/src/backend/bulletproofs.rs:61-80,
// SURVEY.md appendix A.2), fixed-base window tables, and the slot/chunk layouts of each MSM launch.
// One-time initialisation work; no proof is ever computed on the host.
#pragma once
#include <vector>
#include <utility>
#include <cstring>
#include "bp_steps.h"

namespace zkp {

inline void host_sponge(uint8_t* out, size_t outlen, const uint8_t* in, size_t inlen, size_t rate, uint8_t suffix) {
    uint64_t a[25]; memset(a, 0, sizeof a);
    uint8_t* b = reinterpret_cast<uint8_t*>(a);   // little-endian host
    size_t pos = 0;
    for (size_t i = 0; i < inlen; i++) { b[pos++] ^= in[i]; if (pos == rate) { keccak_f1600(a); pos = 0; } }
    b[pos] ^= suffix; b[rate - 1] ^= 0x80;
    keccak_f1600(a);
    size_t done = 0;
    while (done < outlen) {
        size_t n = outlen - done < rate ? outlen - done : rate;
        memcpy(out + done, b, n); done += n;
        if (done < outlen) keccak_f1600(a);
    }
}

inline ge host_from_uniform(const uint8_t b[64]) { uint32_t w[16]; memcpy(w, b, 64); return ge_from_uniform_words(w); }

// ristretto basepoint = edwards25519 basepoint (x even), y = 4/5
inline ge host_basepoint() {
    const uint32_t yw[8] = {0x66666658u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u, 0x66666666u};
    const uint32_t xw[8] = {0x8f25d51au, 0xc9562d60u, 0x9525a7b2u, 0x692cc760u, 0xfdd6dc5cu, 0xc0a4e231u, 0xcd6e53feu, 0x216936d3u};
    ge p; p.X = fe_fromwords(xw); p.Y = fe_fromwords(yw); p.Z = fe_one(); p.T = fe_mul(p.X, p.Y);
    return p;
}

// generators in table order: 0 = B, 1 = B_blinding, 2+i = G_i, 66+i = H_i
inline void host_generators(ge out[NBASE]) {
    out[BASE_B] = host_basepoint();
    uint32_t enc[8]; ge_ristretto_encode(enc, out[BASE_B]);
    uint8_t h[64]; host_sponge(h, 64, reinterpret_cast<uint8_t*>(enc), 32, 72, 0x06);   // SHA3-512
    out[BASE_BB] = host_from_uniform(h);
    for (int which = 0; which < 2; which++) {
        uint8_t in[20]; memcpy(in, "GeneratorsChain", 15);
        in[15] = which ? 'H' : 'G'; in[16] = in[17] = in[18] = in[19] = 0;
        std::vector<uint8_t> stream(64 * BP_N);
        host_sponge(stream.data(), stream.size(), in, 20, 136, 0x1F);                     // SHAKE256
        for (uint32_t i = 0; i < BP_N; i++) out[(which ? BASE_H : BASE_G) + i] = host_from_uniform(stream.data() + 64 * i);
    }
}

inline fe host_fe_invert(const fe& z) {   // z^(p-2) = (z^(2^252-3))^8 * z^3
    fe t = fe_pow22523(z);
    t = fe_sq(fe_sq(fe_sq(t)));
    return fe_mul(t, fe_mul(fe_sq(z), z));
}

// table[(base*NWIN + w)*NENT + e] = (e+1) * 2^(WBITS*w) * Base, affine niels, 30 canonical-limb words each
inline void host_build_table_for_base(uint32_t* dst, const ge& base) {
    std::vector<ge> pts(NWIN * NENT);
    ge pw = base;
    for (uint32_t w = 0; w < NWIN; w++) {
        ge acc = pw;
        for (uint32_t e = 0; e < NENT; e++) { pts[w * NENT + e] = acc; acc = ge_add(acc, pw); }
        for (uint32_t k = 0; k < WBITS; k++) pw = ge_dbl(pw);
    }
    // batch inversion of Z
    const size_t n = pts.size();
    std::vector<fe> pre(n);
    fe run = fe_one();
    for (size_t i = 0; i < n; i++) { pre[i] = run; run = fe_mul(run, pts[i].Z); }
    fe inv = host_fe_invert(run);
    const fe d2 = fe_const_d2();
    for (size_t i = n; i-- > 0;) {
        const fe zi = fe_mul(inv, pre[i]);
        inv = fe_mul(inv, pts[i].Z);
        const fe x = fe_mul(pts[i].X, zi), y = fe_mul(pts[i].Y, zi);
        uint32_t w8[8];
        // store canonical (fully reduced) limbs so that device-side negation 2p - limb never underflows
        fe_towords(w8, fe_add(y, x)); const fe ypx = fe_fromwords(w8);
        fe_towords(w8, fe_sub(y, x)); const fe ymx = fe_fromwords(w8);
        fe_towords(w8, fe_mul(fe_mul(x, y), d2)); const fe xy2d = fe_fromwords(w8);
        uint32_t* q = dst + i * NIELS_W;
        for (int k = 0; k < 10; k++) { q[k] = ypx.v[k]; q[10 + k] = ymx.v[k]; q[20 + k] = xy2d.v[k]; }
    }
}

// ------------------------------------------------------------------------------------------------
struct MsmLayout {
    std::vector<uint16_t> slot_base, chunk_begin, chunk_win0, chunk_nwin, target_chunk_begin;
    std::vector<uint8_t> slot_nwin;
    uint32_t nslots() const { return (uint32_t)slot_base.size(); }
    uint32_t nchunks() const { return (uint32_t)chunk_begin.size(); }
    uint32_t ntargets() const { return (uint32_t)target_chunk_begin.size() - 1; }
};
using SlotList = std::vector<std::pair<uint16_t, uint8_t>>;   // (base, nwin)

// chunks never straddle targets; slot-aligned chunks of at most `win_budget` windows of work
inline MsmLayout make_layout(const std::vector<SlotList>& targets, uint32_t win_budget) {
    MsmLayout L;
    L.target_chunk_begin.push_back(0);
    for (const auto& t : targets) {
        uint32_t used = 0; bool open = false;
        for (const auto& s : t) {
            if (open && used + s.second > win_budget) { L.chunk_nwin.push_back((uint16_t)used); used = 0; open = false; }
            if (!open) { L.chunk_begin.push_back((uint16_t)L.slot_base.size()); L.chunk_win0.push_back(0); open = true; }
            L.slot_base.push_back(s.first); L.slot_nwin.push_back(s.second); used += s.second;
        }
        if (open) L.chunk_nwin.push_back((uint16_t)used);
        L.target_chunk_begin.push_back((uint16_t)L.chunk_begin.size());
    }
    return L;
}
// window-granular chunks: about `total_chunks` chunks of (nearly) equal work, distributed over the targets in proportion
// to their window counts, so that nchunks * (row groups) can be matched to the number of resident workgroups
inline MsmLayout make_layout_even(const std::vector<SlotList>& targets, uint32_t total_chunks) {
    MsmLayout L;
    uint64_t W = 0; for (const auto& t : targets) for (const auto& s : t) W += s.second;
    L.target_chunk_begin.push_back(0);
    for (const auto& t : targets) {
        uint32_t wt = 0; for (const auto& s : t) wt += s.second;
        uint32_t k = (uint32_t)((uint64_t)total_chunks * wt / (W ? W : 1));
        if (k < 1) k = 1;
        if (k > wt && wt) k = wt;
        const uint32_t slot0 = (uint32_t)L.slot_base.size();
        for (const auto& s : t) { L.slot_base.push_back(s.first); L.slot_nwin.push_back(s.second); }
        // walk the flattened (slot, window) sequence and cut it into k ranges
        uint32_t slot = slot0, win = 0, done = 0;
        for (uint32_t c = 0; c < k; c++) {
            const uint32_t upto = (uint32_t)((uint64_t)wt * (c + 1) / k), n = upto - done;
            if (n == 0) continue;
            L.chunk_begin.push_back((uint16_t)slot); L.chunk_win0.push_back((uint16_t)win); L.chunk_nwin.push_back((uint16_t)n);
            uint32_t left = n;
            while (left) { const uint32_t room = L.slot_nwin[slot] - win, step = left < room ? left : room; win += step; left -= step; if (win == L.slot_nwin[slot]) { slot++; win = 0; } }
            done = upto;
        }
        L.target_chunk_begin.push_back((uint16_t)L.chunk_begin.size());
    }
    return L;
}
// The flat step list k_msm_gather walks (msm_kernel.h): for every chunk, in order, one (entry index, digit descriptor) pair per
// (slot, window) step.  slot_scalar: digit row of each slot (nullptr: row = slot); shape of the table: nent entries per window,
// slot_ent entries per point, uneven = the 18-window form of radix 2^14 (window 17 starts one nent later), digw words per digit row.
struct GatherShape { uint32_t nent, slot_ent, uneven, digw; };
inline bool make_gather_steps(const MsmLayout& L, const uint16_t* slot_scalar, const GatherShape& g, std::vector<uint32_t>& steps, std::vector<uint32_t>& chunk_step0) {
    steps.clear(); chunk_step0.clear();
    for (uint32_t c = 0; c < L.nchunks(); c++) {
        chunk_step0.push_back((uint32_t)(steps.size() / 2));
        uint32_t s = L.chunk_begin[c], w = L.chunk_win0[c];
        for (uint32_t left = L.chunk_nwin[c]; left > 0; left--) {
            const uint64_t first = (uint64_t)L.slot_base[s] * g.slot_ent + (uint64_t)(w + (g.uneven && w == 17u ? 1u : 0u)) * g.nent;
            const uint64_t word = (uint64_t)(slot_scalar ? slot_scalar[s] : s) * g.digw + w / 2;
            if (first + g.nent > 0xffffffffull || word > 0x7fffffffull) return false;          // (tables of more than 2^32 entries: not with these circuits)
            steps.push_back((uint32_t)first); steps.push_back((uint32_t)(word << 1) | (w & 1u));
            if (++w == L.slot_nwin[s]) { s++; w = 0; }
        }
    }
    chunk_step0.push_back((uint32_t)(steps.size() / 2));
    for (int pad = 0; pad < 8; pad++) { steps.push_back(0); steps.push_back(0); }              // (the kernel never reads past a chunk's end; slack for wide scalar loads)
    return true;
}

// n = bits per proof (8, 16, 32, 64): party 0's first n generators of each chain (BulletproofGens::new(n, 2)); slot order
// = the digit rows bp_steps.h writes for that n
// nw / nw64: windows of a full-width scalar / of a 64-bit value at the radix of the tables the launch walks (radix 1024: 26 / 7, the
// verifier's LDS-streamed tables; radix 2^16: 16 / 5, the prover's HBM-resident tables of edg.h)
inline std::vector<SlotList> targets_phase1(uint32_t n = BP_N, uint8_t nw = NWIN, uint8_t nw64 = NWIN_U64) {
    const uint8_t NWIN = nw, NWIN_U64 = nw64;
    SlotList v = {{BASE_B, NWIN_U64}, {BASE_BB, NWIN}}, a = {{BASE_BB, NWIN}}, s = {{BASE_BB, NWIN}};
    for (uint32_t i = 0; i < n; i++) a.push_back({(uint16_t)(BASE_G + i), 1});
    for (uint32_t i = 0; i < n; i++) a.push_back({(uint16_t)(BASE_H + i), 1});
    for (uint32_t i = 0; i < n; i++) s.push_back({(uint16_t)(BASE_G + i), NWIN});
    for (uint32_t i = 0; i < n; i++) s.push_back({(uint16_t)(BASE_H + i), NWIN});
    return {v, a, s};
}
inline std::vector<SlotList> targets_phase2(uint8_t nw = NWIN) { SlotList t = {{BASE_B, nw}, {BASE_BB, nw}}; return {t, t}; }
inline std::vector<SlotList> targets_round(uint32_t r, uint32_t n = BP_N, uint8_t nw = NWIN) {
    const uint8_t NWIN = nw;
    uint32_t lg = 0; while ((1u << lg) < n) lg++;
    const uint32_t p = lg - 1 - r, k = 1u << p, half = n / 2;
    auto idx = [&](uint32_t rank, uint32_t bit) { return ((rank >> p) << (p + 1)) | (bit << p) | (rank & (k - 1)); };
    SlotList l = {{BASE_B, NWIN}}, rr = {{BASE_B, NWIN}};
    for (uint32_t q = 0; q < half; q++) l.push_back({(uint16_t)(BASE_G + idx(q, 1)), NWIN});
    for (uint32_t q = 0; q < half; q++) l.push_back({(uint16_t)(BASE_H + idx(q, 0)), NWIN});
    for (uint32_t q = 0; q < half; q++) rr.push_back({(uint16_t)(BASE_G + idx(q, 0)), NWIN});
    for (uint32_t q = 0; q < half; q++) rr.push_back({(uint16_t)(BASE_H + idx(q, 1)), NWIN});
    return {l, rr};
}
inline std::vector<SlotList> targets_ctask(uint8_t nw = NWIN, uint8_t nw64 = NWIN_U64) { return {SlotList{{BASE_B, nw64}, {BASE_BB, nw}}}; }
// range-proof verification: one target over all 130 generators (slot index == generator index == digit row)
inline std::vector<SlotList> targets_verify(uint8_t nw = NWIN) {
    const uint8_t NWIN = nw;
    SlotList t = {{BASE_B, NWIN}, {BASE_BB, NWIN}};
    for (uint32_t i = 0; i < BP_N; i++) t.push_back({(uint16_t)(BASE_G + i), NWIN});
    for (uint32_t i = 0; i < BP_N; i++) t.push_back({(uint16_t)(BASE_H + i), NWIN});
    return {t};
}
inline MsmLayout layout_phase1(uint32_t budget, uint32_t n = BP_N) { return make_layout(targets_phase1(n), budget); }
inline MsmLayout layout_phase2(uint32_t budget) { return make_layout(targets_phase2(), budget); }
inline MsmLayout layout_round(uint32_t r, uint32_t budget, uint32_t n = BP_N) { return make_layout(targets_round(r, n), budget); }
inline MsmLayout layout_ctask(uint32_t budget) { return make_layout(targets_ctask(), budget); }

}  // namespace zkp
