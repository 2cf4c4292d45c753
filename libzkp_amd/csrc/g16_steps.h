// Per-thread / per-workgroup steps of the batched Groth16 prover over BN254 for libzkp's two circuits
// (EqualityCircuit /root/reference/src/backend/snark.rs:262-291, MembershipCircuit :514-585), i.e. what
// Groth16::<Bn254>::prove does at snark.rs:364,442 (ark-groth16 create_proof_with_reduction + LibsnarkReduction
// witness map; restated in oracle/py/groth16.py and SURVEY.md appendix A.4).
//
// MI355X-first structure: one lane = one proof for witness generation and final assembly; one workgroup = one proof
// for the QAP division (all seven size-m NTTs of a proof stay in LDS); every MSM is a fixed-base, lane = proof,
// LDS-streamed-window kernel over the proving key's points (same kernel template as the Bulletproofs path).
#pragma once
#include "bn254_g.h"
#include "bn254_fr9.h"
#include "keccak.h"
#include "sc25519.h"

namespace zkp {

// Key-point tables: signed radix-2^w digits (two 16-bit digits per word), ceil(255 / w) windows of 2^(w-1) affine entries per key
// point, resident in HBM and gathered per lane (k_msm_gather).  The radix is a property of a LOADED KEY, chosen when the key is
// installed from the memory the GPU has free (g16_impl.inc: 2^14 when both circuits' ~65 GB fit -- 19 mixed additions per 254-bit
// scalar, 5 per 64-bit value -- then 2^13, 2^12, ... down to 2^8, 33 MB per circuit); ZKP_HIP_G16_WBITS forces one.  Measured on the
// 4096-op mixed batch in round 2: radix 2^13 16.0 ms, 2^14 15.6 ms, 2^15 another ~1.5 % for 143 GB; the LDS-streamed radix-1024
// tables of round 1 needed 26 additions per scalar.
#ifndef ZKP_G16_WBITS
#define ZKP_G16_WBITS 14
#endif
// Radix of a key's window tables.  Even form: nwin windows of wbits bits, nent = 2^(wbits-1) entries each (signed digits).  Uneven form
// (wbits = 14 only): a 254-bit scalar in 18 windows instead of 19 -- at 19 x 14 bits the last window holds two bits and four of its 8192
// entries are ever read -- as sixteen 14-bit windows, one 15-bit window (signed digits, 2 nent entries) and the top 15 bits (bits 239..253 of
// a scalar < r: unsigned digits <= 24 784, no carry leaves it): one addition fewer per full-width scalar for 172 544 instead of 155 648
// entries per base.  Scalars of the short classes (64-bit values, bits) only touch the 14-bit windows.
struct G16Radix { uint32_t wbits, nwin, nent, digw, nwin_u64, uneven, slot_ent; };
constexpr uint32_t G16_UNEVEN_TOP_ENT = 25088;          // 49 segments of 512 >= 24 784
ZKP_HD inline G16Radix g16_radix(uint32_t wbits, bool uneven = false) {
    G16Radix r; r.wbits = wbits; r.nwin = (254 + wbits) / wbits; r.nent = 1u << (wbits - 1); r.nwin_u64 = (64 + wbits) / wbits; r.uneven = 0;
    r.slot_ent = r.nwin * r.nent;       // nwin * wbits >= 255 and nwin_u64 * wbits >= 65: the windows cover the scalar and the recoding carry
    if (uneven && wbits == 14) { r.uneven = 1; r.nwin = 18; r.slot_ent = 18u * r.nent + G16_UNEVEN_TOP_ENT; }
    r.digw = (r.nwin + 1) / 2;
    return r;
}
ZKP_HD inline uint32_t g16_win_bit(const G16Radix& rx, uint32_t w) { return rx.uneven && w == 17 ? 239u : rx.wbits * w; }      // first bit of window w
ZKP_HD inline uint32_t g16_win_off(const G16Radix& rx, uint32_t w) { return (w + (rx.uneven && w == 17 ? 1u : 0u)) * rx.nent; }  // first entry of window w in a base's block
ZKP_HD inline uint32_t g16_win_ent(const G16Radix& rx, uint32_t w) { return !rx.uneven || w < 16 ? rx.nent : w == 16 ? 2u * rx.nent : G16_UNEVEN_TOP_ENT; }
constexpr uint32_t G16_WBITS_DEFAULT = ZKP_G16_WBITS, G16_WBITS_KNEE = 13, G16_WBITS_MIN = 8, G16_WBITS_MAX = 15, G16_DIGW_MAX = 16;      // radix 2^8: 32 windows = 16 digit words
static_assert(G16_WBITS_DEFAULT >= G16_WBITS_MIN && G16_WBITS_DEFAULT <= G16_WBITS_MAX, "key-table radix out of range");
// packed signed digits of a raw canonical scalar at the radix rx (packed[] holds G16_DIGW_MAX words; the first rx.digw are meaningful)
ZKP_HD inline void g16_recode_uneven(uint32_t* packed, const sc& raw) {
    uint32_t carry = 0;
    ZKP_UNROLL for (int j = 0; j < 18; j++) {
        const int bit = j == 17 ? 239 : 14 * j, wb = j < 16 ? 14 : 15, wd = bit >> 5, sh = bit & 31;
        uint32_t x = wd < 8 ? raw.v[wd] >> sh : 0u;
        if (sh + wb > 32 && wd + 1 < 8) x |= raw.v[wd + 1] << (32 - sh);
        uint32_t d = (x & ((1u << wb) - 1u)) + carry;
        carry = j < 17 && d > (1u << (wb - 1)) ? 1u : 0u;          // the top window keeps its digit unsigned: nothing above it takes a carry
        d = (d - (carry << wb)) & 0xffffu;
        if ((j & 1) == 0) packed[j >> 1] = d; else packed[j >> 1] |= d << 16;
    }
}
ZKP_HD inline void g16_recode(uint32_t* packed, const sc& raw, const G16Radix& rx) {
    if (rx.uneven) { g16_recode_uneven(packed, raw); return; }
    switch (rx.wbits) {
        case 8: sc_recode_signed<8, 32>(packed, raw); break;
        case 9: sc_recode_signed<9, 29>(packed, raw); break;
        case 10: sc_recode_signed<10, 26>(packed, raw); break;
        case 11: sc_recode_signed<11, 24>(packed, raw); break;
        case 12: sc_recode_signed<12, 22>(packed, raw); break;
        case 13: sc_recode_signed<13, 20>(packed, raw); break;
        case 14: sc_recode_signed<14, 19>(packed, raw); break;
        default: sc_recode_signed<15, 17>(packed, raw); break;
    }
}
// the verifier's public-input tables (gamma_abc_g1: a hundred-odd points, 64-bit scalars mostly) keep a fixed small radix
constexpr uint32_t G16V_WBITS = 10, G16V_NWIN = 26, G16V_NENT = 512, G16V_NWIN_U64 = 7;

constexpr uint32_t MIMC_ROUNDS = 110, G16_MAX_SET = 64;
constexpr uint32_t G16_TAPE_IDX = 0x47313600u;
enum { G16_EQUALITY = 0, G16_MEMBERSHIP = 1 };

ZKP_HD inline fr ld_fr(const uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows) {
    fr r; const uint32_t* q = p + (size_t)idx * 8 * rows + row;
    ZKP_UNROLL for (int k = 0; k < 8; k++) r.v[k] = q[(size_t)k * rows];
    return r;
}
ZKP_HD inline void st_fr(uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows, const fr& s) {
    uint32_t* q = p + (size_t)idx * 8 * rows + row;
    ZKP_UNROLL for (int k = 0; k < 8; k++) q[(size_t)k * rows] = s.v[k];
}
ZKP_HD inline fr ld_fr_c(const uint32_t* p, uint32_t idx) { fr r; ZKP_UNROLL for (int k = 0; k < 8; k++) r.v[k] = p[(size_t)idx * 8 + k]; return r; }
ZKP_HD inline void g16_put_bytes(uint8_t* dst, const uint32_t* w, int nwords) {
    for (int i = 0; i < nwords; i++) { dst[4 * i] = (uint8_t)w[i]; dst[4 * i + 1] = (uint8_t)(w[i] >> 8); dst[4 * i + 2] = (uint8_t)(w[i] >> 16); dst[4 * i + 3] = (uint8_t)(w[i] >> 24); }
}
// packed signed digits of a Montgomery-form Fr element (canonical value < r < 2^254)
ZKP_HD inline void st_fr_digits(uint32_t* d, uint32_t idx, uint32_t row, uint32_t rows, const fr& x, const G16Radix& rx) {
    sc raw; fp_to_raw(raw.v, x);
    uint32_t pk[G16_DIGW_MAX]; g16_recode(pk, raw, rx);
    uint32_t* q = d + (size_t)idx * rx.digw * rows + row;
    ZKP_UNROLL for (uint32_t k = 0; k < G16_DIGW_MAX; k++) if (k < rx.digw) q[(size_t)k * rows] = pk[k];
}

struct G16View {
    uint32_t rows, kind;
    G16Radix rx;                              // radix of the loaded key's tables = of every digit row below
    uint32_t n_inst, n_wit, nv, m;            // circuit shape; m = domain size
    // inputs
    const uint64_t* value;                    // [rows] the committed value (a for equality, value for membership)
    const uint64_t* set_vals;                 // membership: [rows][64] padded set, else null
    const uint32_t* set_len;                  // membership: [rows]
    const uint32_t* seeds;                    // [rows][8]
    const uint32_t* mimc_c;                   // [110][8] round constants (Montgomery Fr)
    // workspace
    uint32_t* z;                              // [nv][8][rows] full assignment (instance block first), Montgomery Fr
    uint32_t* sdig;                           // [nscalars][rx.digw][rows] packed digits: z_k (nv), h_i (m-1), r, s, -rs, one
    uint32_t* rs;                             // [2][8][rows] raw canonical r, s (for the variable-base part of C)
    uint32_t* qap_evals;                      // [rows][2][9][m] the QAP step's coset evaluations of a and b while c is transformed
    // output
    uint8_t* out; uint64_t stride;            // envelope per row
};
ZKP_HD inline uint32_t g16_sc_h(const G16View& V) { return V.nv; }
ZKP_HD inline uint32_t g16_sc_r(const G16View& V) { return V.nv + V.m - 1; }
ZKP_HD inline uint32_t g16_sc_s(const G16View& V) { return V.nv + V.m; }
ZKP_HD inline uint32_t g16_sc_nrs(const G16View& V) { return V.nv + V.m + 1; }
ZKP_HD inline uint32_t g16_sc_one(const G16View& V) { return V.nv + V.m + 2; }
ZKP_HD inline uint32_t g16_nscalars(uint32_t nv, uint32_t m) { return nv + m + 3; }

// MiMC-5 chain (snark.rs:201-211) writing the circuit's intermediate witnesses t^2, t^4, t^5 (snark.rs:232-247).
// wit0 = index in z of the first MiMC witness.  Returns the hash.
ZKP_HD inline fr g16_mimc_chain(const G16View& V, uint32_t row, fr x, uint32_t wit0) {
    for (uint32_t i = 0; i < MIMC_ROUNDS; i++) {
        const fr t = fp_add(x, ld_fr_c(V.mimc_c, i));
        const fr t2 = fp_sq(t), t4 = fp_sq(t2);
        x = fp_mul(t4, t);
        if (V.z) { st_fr(V.z, wit0 + 3 * i, row, V.rows, t2); st_fr(V.z, wit0 + 3 * i + 1, row, V.rows, t4); st_fr(V.z, wit0 + 3 * i + 2, row, V.rows, x); }
    }
    return x;
}

// thread = proof.  Fills z, the envelope header + commitment, r/s and their digits.
ZKP_HD inline void step_g16_witness(const G16View& V, uint32_t row) {
    const uint32_t rows = V.rows;
    const uint64_t val = V.value[row];
    const fr one = fp_one<FrParams>(), zero = fp_zero<FrParams>();
    const fr v = fp_from_u64<FrParams>(val);
    uint8_t* o = V.out + (uint64_t)row * V.stride;
    st_fr(V.z, 0, row, rows, one);
    fr h;
    uint32_t proof_len = 256;
    if (V.kind == G16_EQUALITY) {
        // instance: [1, commitment]; witness: a, b, MiMC(3 x 110)          (snark.rs:262-291)
        st_fr(V.z, 2, row, rows, v); st_fr(V.z, 3, row, rows, v);
        h = g16_mimc_chain(V, row, v, 4);
        st_fr(V.z, 1, row, rows, h);
        o[0] = 2; o[1] = 2;
    } else {
        // instance: [1, commitment, set[64], is_real[64]]; witness: value, MiMC, sel[64], sel*(1-is_real)[64], sel*(value-set)[64]
        const uint32_t len = V.set_len[row];
        const uint64_t* sv = V.set_vals + (size_t)row * G16_MAX_SET;
        const uint32_t w0 = V.n_inst;                       // first witness index
        st_fr(V.z, w0, row, rows, v);
        h = g16_mimc_chain(V, row, v, w0 + 1);
        st_fr(V.z, 1, row, rows, h);
        uint32_t pos = 0xffffffffu;
        for (uint32_t i = 0; i < len; i++) if (pos == 0xffffffffu && sv[i] == val) pos = i;     // set.iter().position (snark.rs:415-418)
        const uint32_t sel0 = w0 + 1 + 3 * MIMC_ROUNDS;
        for (uint32_t i = 0; i < G16_MAX_SET; i++) {
            const bool real = i < len;
            const fr si = fp_from_u64<FrParams>(real ? sv[i] : 0);
            st_fr(V.z, 2 + i, row, rows, si);
            st_fr(V.z, 2 + G16_MAX_SET + i, row, rows, real ? one : zero);
            const bool sel = i == pos;
            st_fr(V.z, sel0 + i, row, rows, sel ? one : zero);
            st_fr(V.z, sel0 + G16_MAX_SET + i, row, rows, (sel && !real) ? one : zero);        // sel * (1 - is_real)
            st_fr(V.z, sel0 + 2 * G16_MAX_SET + i, row, rows, sel ? fp_sub(v, si) : zero);       // sel * (value - set_i)
        }
        o[0] = 2; o[1] = 4;
        proof_len = 4 + 8 * len + 256;
        uint8_t* pl = o + 10;
        pl[0] = (uint8_t)len; pl[1] = (uint8_t)(len >> 8); pl[2] = (uint8_t)(len >> 16); pl[3] = (uint8_t)(len >> 24);
        for (uint32_t i = 0; i < len; i++) for (int k = 0; k < 8; k++) pl[4 + 8 * i + k] = (uint8_t)(sv[i] >> (8 * k));
    }
    for (int k = 0; k < 4; k++) { o[2 + k] = (uint8_t)(proof_len >> (8 * k)); o[6 + k] = (uint8_t)(32u >> (8 * k)); }
    uint32_t hraw[8]; fp_to_raw(hraw, h);
    g16_put_bytes(o + 10 + proof_len, hraw, 8);          // commitment = fr_to_commitment(mimc) (commitment.rs:14-16)
    // prover randomness
    uint32_t seed[8]; ZKP_UNROLL for (int k = 0; k < 8; k++) seed[k] = V.seeds[(size_t)row * 8 + k];
    uint32_t w[16];
    tape_draw64(w, seed, G16_TAPE_IDX, 0); const fr r = fp_from_wide<FrParams>(w);
    tape_draw64(w, seed, G16_TAPE_IDX, 1); const fr s = fp_from_wide<FrParams>(w);
    st_fr_digits(V.sdig, g16_sc_r(V), row, rows, r, V.rx);
    st_fr_digits(V.sdig, g16_sc_s(V), row, rows, s, V.rx);
    st_fr_digits(V.sdig, g16_sc_nrs(V), row, rows, fp_neg(fp_mul(r, s)), V.rx);
    uint32_t* q = V.sdig + (size_t)g16_sc_one(V) * V.rx.digw * rows + row;
    q[0] = 1u; for (uint32_t k = 1; k < V.rx.digw; k++) q[(size_t)k * rows] = 0u;
    fr rr, sr; fp_to_raw(rr.v, r); fp_to_raw(sr.v, s);
    st_fr(V.rs, 0, row, rows, rr); st_fr(V.rs, 1, row, rows, sr);
}
// thread = (variable k, proof): digits of z_k
ZKP_HD inline void step_g16_zdigits(const G16View& V, uint32_t k, uint32_t row) {
    st_fr_digits(V.sdig, k, row, V.rows, ld_fr(V.z, k, row, V.rows), V.rx);
}

// ---- R1CS matrices in CSR form (coefficients in Montgomery Fr) and the domain tables, all read-only
struct G16Circuit {
    uint32_t n_rows, m, logm;
    const uint32_t *a_ptr, *a_col, *a_coef, *b_ptr, *b_col, *b_coef, *c_ptr, *c_col, *c_coef;
    const uint32_t *tw, *tw_inv;         // [m/2][8]  w^j, w^-j
    const uint32_t *coset, *coset_inv;   // [m][8]    g^i / m ,  g^-i / m
    const uint32_t* zinv;                // [8]       1 / (g^m - 1)
};
ZKP_HD inline fr g16_row_dot(const uint32_t* ptr, const uint32_t* col, const uint32_t* coef, uint32_t r, const uint32_t* z, uint32_t row, uint32_t rows) {
    fr acc = fp_zero<FrParams>();
    for (uint32_t e = ptr[r]; e < ptr[r + 1]; e++) acc = fp_add(acc, fp_mul(ld_fr_c(coef, e), ld_fr(z, col[e], row, rows)));
    return acc;
}
ZKP_HD inline uint32_t g16_bitrev(uint32_t x, uint32_t bits) {
    uint32_t r = 0; for (uint32_t i = 0; i < bits; i++) { r = (r << 1) | (x & 1u); x >>= 1; } return r;
}

// The QAP witness map of ONE proof (LibsnarkReduction::witness_map_from_matrices), `nthreads` lanes cooperating, `sync` the workgroup barrier.
//
// Round 4: ONE polynomial in LDS at a time, every pass of a transform does TWO butterfly stages, m / 4 lanes per proof.
// Rounds 1-3 kept a, b and c in LDS together (54 / 108 KB: two / one workgroup per CU) and ran one radix-2 stage per barrier on 512 lanes
// -- half of them idle at m = 512 --, 27 / 30 barriers per proof with two waves per SIMD to cover them: VALU active 28 %, time in the
// barriers.  Now a workgroup of m / 4 lanes (128 / 256: two / four waves) holds nine limb rows of ONE polynomial (18 / 36 KB: eight / four
// workgroups per CU, whose barriers interleave), a lane loads four elements, does the two butterflies of one stage and the two of the
// next in registers and stores them: half the LDS traffic and half the barriers per stage.  a and b leave their coset evaluations in
// a per-proof scratch block in HBM (2 x 36 m bytes, written and read once, coalesced) while c is transformed; the pointwise step reads
// them back.  The arithmetic of every butterfly is unchanged (same operations in the same order on the same values: the bounds of
// bn254_fr9.h / tests/test_fq_bounds.py hold as they are).
//
// Image: word row k (limb k of every element) is m words; element e of a row sits at sw(e) = e ^ (bit 5 of e ? 01010b : 0) ^ (bit 6 of e ?
// 10101b : 0).  In a fused pass over stages (h, 2h) lane q touches e = (q / h) 4h + q % h + c h (c = 0..3); for h < 32 the 32 lanes of a
// half-wave spread over address bits 5 and 6 while bits inside the low five go unused, and the two XOR masks fold those two bits back
// into the unused bank bits for every h = 1, 2, 4, 8, 16 (the five 5 x 5 bit matrices are each of full rank), so every ds_read_b32 /
// ds_write_b32 of a pass hits 32 different banks; within an aligned block of 32 elements sw() is a permutation, so the passes that walk
// the image in element order (load, scale, pointwise, store) stay conflict-free as well.
struct G16Lds {
    uint32_t* base; uint32_t m;
    ZKP_HD static uint32_t sw(uint32_t e) { return e ^ ((0u - ((e >> 5) & 1u)) & 10u) ^ ((0u - ((e >> 6) & 1u)) & 21u); }
    ZKP_HD fr9 ld(uint32_t e) const { fr9 r; const uint32_t p = sw(e); ZKP_UNROLL for (int k = 0; k < 9; k++) r.v[k] = base[(size_t)k * m + p]; return r; }
    ZKP_HD void st(uint32_t e, const fr9& v) const { const uint32_t p = sw(e); ZKP_UNROLL for (int k = 0; k < 9; k++) base[(size_t)k * m + p] = v.v[k]; }
};
ZKP_HD inline fr9 ld_fr9_c(const uint32_t* p, uint32_t idx) { fr9 r; ZKP_UNROLL for (int k = 0; k < 9; k++) r.v[k] = p[(size_t)idx * 9 + k]; return r; }
// <row j of one R1CS matrix, z>.  Most coefficients of these circuits are 1 (linear combinations of MiMC rounds, the selector sums)
// and many entries multiply the constant-one variable z_0: neither needs a field multiplication.
ZKP_HD inline fr g16_row_dot_fast(const uint32_t* ptr, const uint32_t* col, const uint32_t* coef, uint32_t r, const uint32_t* z, uint32_t row, uint32_t rows) {
    const fr one = fp_one<FrParams>();
    fr acc = fp_zero<FrParams>();
    for (uint32_t e = ptr[r]; e < ptr[r + 1]; e++) {
        const fr cf = ld_fr_c(coef, e);
        const uint32_t c = col[e];
        bool is_one = true; ZKP_UNROLL for (int k = 0; k < 8; k++) is_one = is_one && cf.v[k] == one.v[k];
        if (c == 0) acc = fp_add(acc, cf);                             // z_0 = 1 (the instance's leading one; Montgomery one times cf)
        else if (is_one) acc = fp_add(acc, ld_fr(z, c, row, rows));
        else acc = fp_add(acc, fp_mul(cf, ld_fr(z, c, row, rows)));
    }
    return acc;
}
// evaluations of polynomial `poly` (0 a, 1 b, 2 c) on the domain, natural order: element j = <row j, z>; a also carries the instance
// rows (a[n_constraints + i] = z_i)
ZKP_HD inline void g16_qap_load(const G16View& V, const G16Circuit& C, const G16Lds& L, uint32_t poly, uint32_t row, uint32_t tid, uint32_t nthreads) {
    const uint32_t nc = C.n_rows;
    const uint32_t *ptr = poly == 0 ? C.a_ptr : poly == 1 ? C.b_ptr : C.c_ptr, *col = poly == 0 ? C.a_col : poly == 1 ? C.b_col : C.c_col,
                   *coef = poly == 0 ? C.a_coef : poly == 1 ? C.b_coef : C.c_coef;
    for (uint32_t j = tid; j < C.m; j += nthreads) {
        fr a = fp_zero<FrParams>();
        if (j < nc) a = g16_row_dot_fast(ptr, col, coef, j, V.z, row, V.rows);
        else if (poly == 0 && j < nc + V.n_inst) a = ld_fr(V.z, j - nc, row, V.rows);
        L.st(j, fr9_from_fr(a));                                        // < 1.4 r
    }
}
// the two butterflies (bounds: bn254_fr9.h)
ZKP_HD inline void g16_bf_dif(fr9& u, fr9& v, const fr9& w) { const fr9 s = fr9_reduce_weak(fr9_add(u, v)); v = fr9_mul(fr9_sub_k<4>(u, v), w); u = s; }
ZKP_HD inline void g16_bf_dit(fr9& u, fr9& v, const fr9& w) { const fr9 t = fr9_mul(v, w); v = fr9_sub_k<2>(u, t); u = fr9_add(u, t); }
// one stage on its own (the odd stage of m = 512): m / 2 butterflies of half-length `half`
template <bool DIF>
ZKP_HD inline void g16_stage_single(const G16Circuit& C, const G16Lds& L, const uint32_t* tw, uint32_t half, uint32_t tid, uint32_t nthreads) {
    const uint32_t len = 2 * half, tstride = C.m / len;
    for (uint32_t k = tid; k < C.m / 2; k += nthreads) {
        const uint32_t grp = k / half, j = k % half, i0 = grp * len + j, i1 = i0 + half;
        const fr9 w = ld_fr9_c(tw, j * tstride);
        fr9 u = L.ld(i0), v = L.ld(i1);
        if (DIF) g16_bf_dif(u, v, w); else g16_bf_dit(u, v, w);
        L.st(i0, u); L.st(i1, v);
    }
}
// two stages in one pass: half-lengths (2h, h) for decimation in frequency, (h, 2h) for decimation in time; m / 4 tasks
template <bool DIF>
ZKP_HD inline void g16_stage_pair(const G16Circuit& C, const G16Lds& L, const uint32_t* tw, uint32_t h, uint32_t tid, uint32_t nthreads) {
    const uint32_t s1 = C.m / (2 * h), s2 = C.m / (4 * h);           // twiddle strides of the stages with half-length h and 2h
    for (uint32_t q = tid; q < C.m / 4; q += nthreads) {
        const uint32_t grp = q / h, j = q % h, e0 = grp * 4 * h + j;
        fr9 x0 = L.ld(e0), x1 = L.ld(e0 + h), x2 = L.ld(e0 + 2 * h), x3 = L.ld(e0 + 3 * h);
        const fr9 wh = ld_fr9_c(tw, j * s1), wa = ld_fr9_c(tw, j * s2), wb = ld_fr9_c(tw, (j + h) * s2);
        if (DIF) { g16_bf_dif(x0, x2, wa); g16_bf_dif(x1, x3, wb); g16_bf_dif(x0, x1, wh); g16_bf_dif(x2, x3, wh); }
        else { g16_bf_dit(x0, x1, wh); g16_bf_dit(x2, x3, wh); g16_bf_dit(x0, x2, wa); g16_bf_dit(x1, x3, wb); }
        L.st(e0, x0); L.st(e0 + h, x1); L.st(e0 + 2 * h, x2); L.st(e0 + 3 * h, x3);
    }
}
// decimation in frequency, natural in, bit-reversed out: half-lengths m/2 ... 1
template <class Sync> ZKP_HD inline void g16_transform_dif(const G16Circuit& C, const G16Lds& L, const uint32_t* tw, uint32_t tid, uint32_t nthreads, Sync sync) {
    uint32_t half = C.m / 2;
    if (C.logm & 1u) { g16_stage_single<true>(C, L, tw, half, tid, nthreads); sync(); half >>= 1; }
    for (; half >= 2; half >>= 2) { g16_stage_pair<true>(C, L, tw, half / 2, tid, nthreads); sync(); }
}
// decimation in time, bit-reversed in, natural out: half-lengths 1 ... m/2
template <class Sync> ZKP_HD inline void g16_transform_dit(const G16Circuit& C, const G16Lds& L, const uint32_t* tw, uint32_t tid, uint32_t nthreads, Sync sync) {
    uint32_t h = 1;
    for (uint32_t i = 0; i < C.logm / 2; i++, h <<= 2) { g16_stage_pair<false>(C, L, tw, h, tid, nthreads); sync(); }
    if (C.logm & 1u) { g16_stage_single<false>(C, L, tw, C.m / 2, tid, nthreads); sync(); }
}
// element at position p of a bit-reversed image has index bitrev(p): scale it by tab[bitrev(p)]
ZKP_HD inline void g16_scale_bitrev(const G16Circuit& C, const G16Lds& L, const uint32_t* tab, uint32_t tid, uint32_t nthreads) {
    for (uint32_t p = tid; p < C.m; p += nthreads) L.st(p, fr9_mul(L.ld(p), ld_fr9_c(tab, g16_bitrev(p, C.logm))));
}
// the image as it lies in LDS <-> the proof's scratch block (coalesced both ways; the swizzle travels with the words)
ZKP_HD inline void g16_spill(const G16Circuit& C, const G16Lds& L, uint32_t* dst, uint32_t tid, uint32_t nthreads) {
    for (uint32_t w = tid; w < 9 * C.m; w += nthreads) dst[w] = L.base[w];
}
ZKP_HD inline fr9 g16_ld_spilled(const uint32_t* src, uint32_t m, uint32_t e) { fr9 r; const uint32_t p = G16Lds::sw(e); ZKP_UNROLL for (int k = 0; k < 9; k++) r.v[k] = src[(size_t)k * m + p]; return r; }
// (a b - c) / Z(g w^j) with a, b from the scratch block and c in LDS
ZKP_HD inline void g16_pointwise(const G16Circuit& C, const G16Lds& L, const uint32_t* ea, const uint32_t* eb, uint32_t tid, uint32_t nthreads) {
    const fr9 zinv = ld_fr9_c(C.zinv, 0);
    for (uint32_t i = tid; i < C.m; i += nthreads)          // operands are outputs of ten decimation-in-time stages (< 22 r): a b < 3.9 r, c < 32 r
        L.st(i, fr9_mul(fr9_sub_k<32>(fr9_mul(g16_ld_spilled(ea, C.m, i), g16_ld_spilled(eb, C.m, i)), L.ld(i)), zinv));
}
ZKP_HD inline void g16_store_h_bitrev(const G16View& V, const G16Circuit& C, const G16Lds& L, uint32_t row, uint32_t tid, uint32_t nthreads) {
    for (uint32_t p = tid; p < C.m; p += nthreads) {
        const uint32_t i = g16_bitrev(p, C.logm);             // coefficient index of the element at position p
        if (i + 1 < C.m) st_fr_digits(V.sdig, g16_sc_h(V) + i, row, V.rows, fr9_to_fr<FrParams>(fr9_mul(L.ld(p), ld_fr9_c(C.coset_inv, i))), V.rx);
    }
}
// the whole witness map, written so that host emulation (nthreads = 1, sync = no-op) and the kernel share it.  Per polynomial: DIF
// inverse (natural -> bit-reversed), coset scaling by index, DIT forward (bit-reversed -> natural); then pointwise, DIF inverse,
// coefficients read off by index: no pass of the LDS image is a bit-reversal scatter.  scratch: 2 x 9 x m words of this proof.
template <class Sync>
ZKP_HD inline void g16_qap_proof(const G16View& V, const G16Circuit& C, const G16Lds& L, uint32_t* scratch, uint32_t row, uint32_t tid, uint32_t nthreads, Sync sync) {
    for (uint32_t poly = 0; poly < 3; poly++) {
        g16_qap_load(V, C, L, poly, row, tid, nthreads); sync();
        g16_transform_dif(C, L, C.tw_inv, tid, nthreads, sync);                              // iFFT (x m), bit-reversed out
        g16_scale_bitrev(C, L, C.coset, tid, nthreads); sync();                              // /m and coset shift g^i
        g16_transform_dit(C, L, C.tw, tid, nthreads, sync);                                  // coset FFT, natural out
        if (poly < 2) { g16_spill(C, L, scratch + (size_t)poly * 9 * C.m, tid, nthreads); sync(); }
    }
    g16_pointwise(C, L, scratch, scratch + (size_t)9 * C.m, tid, nthreads); sync();          // (a*b - c) / Z(g w^j)
    g16_transform_dif(C, L, C.tw_inv, tid, nthreads, sync);                                  // coset iFFT, bit-reversed out
    g16_store_h_bitrev(V, C, L, row, tid, nthreads);                                         // /m, g^-i, digits
}

// ---- final assembly.  thread = proof.  sums: [4 targets][words][rows] Jacobian sums A (G1), B1 (G1), Cp (G1), B2 (G2)
// Jacobian points in global memory: ten 26-bit limbs per Fq coordinate, word-major ([word][row]) so that lane = row is coalesced
#define G1_JAC_W 30
#define G2_JAC_W 60
ZKP_HD inline fq ld_fq_col(const uint32_t* q, uint32_t rows) { fq r; ZKP_UNROLL for (int k = 0; k < 10; k++) r.v[k] = q[(size_t)k * rows]; return r; }
ZKP_HD inline void st_fq_col(uint32_t* q, uint32_t rows, const fq& a) { ZKP_UNROLL for (int k = 0; k < 10; k++) q[(size_t)k * rows] = a.v[k]; }
ZKP_HD inline g1_jac ld_g1_jac(const uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows) {
    const uint32_t* q = p + (size_t)idx * G1_JAC_W * rows + row; const size_t s = (size_t)10 * rows;
    g1_jac r; r.X = ld_fq_col(q, rows); r.Y = ld_fq_col(q + s, rows); r.Z = ld_fq_col(q + 2 * s, rows);
    return r;
}
ZKP_HD inline void st_g1_jac(uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows, const g1_jac& g) {
    uint32_t* q = p + (size_t)idx * G1_JAC_W * rows + row; const size_t s = (size_t)10 * rows;
    st_fq_col(q, rows, g.X); st_fq_col(q + s, rows, g.Y); st_fq_col(q + 2 * s, rows, g.Z);
}
ZKP_HD inline g2_jac ld_g2_jac(const uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows) {
    const uint32_t* q = p + (size_t)idx * G2_JAC_W * rows + row; const size_t s = (size_t)10 * rows;
    g2_jac r;
    r.X.c0 = ld_fq_col(q, rows); r.X.c1 = ld_fq_col(q + s, rows); r.Y.c0 = ld_fq_col(q + 2 * s, rows); r.Y.c1 = ld_fq_col(q + 3 * s, rows);
    r.Z.c0 = ld_fq_col(q + 4 * s, rows); r.Z.c1 = ld_fq_col(q + 5 * s, rows);
    return r;
}
ZKP_HD inline void st_g2_jac(uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows, const g2_jac& g) {
    uint32_t* q = p + (size_t)idx * G2_JAC_W * rows + row; const size_t s = (size_t)10 * rows;
    st_fq_col(q, rows, g.X.c0); st_fq_col(q + s, rows, g.X.c1); st_fq_col(q + 2 * s, rows, g.Y.c0); st_fq_col(q + 3 * s, rows, g.Y.c1);
    st_fq_col(q + 4 * s, rows, g.Z.c0); st_fq_col(q + 5 * s, rows, g.Z.c1);
}
// C = Cp + s*A + r*B1 (Cp already contains l_aux, h and -rs*delta); proof = A || B2 || C  (snark.rs:369-373).
// Split over lanes: thread (part, row) of step_g16_cparts computes one GLV half of s*A (part 0, 1) or of r*B1 (part 2, 3):
// k P = k1 P + k2 phi(P) with half-length k1, k2 (bn254_g.h), so the chain a lone wave has to walk is 132 doublings instead
// of 256; thread (which, row) of step_g16_final serialises A (0), B2 (1) or assembles and serialises C (2).
constexpr uint32_t G16_CPARTS = 4;
ZKP_HD inline void step_g16_cparts(const G16View& V, const uint32_t* sum_g1, uint32_t* tmp_g1, uint32_t part, uint32_t row) {
    const uint32_t rows = V.rows, which = part >> 1;
    g1_jac P = ld_g1_jac(sum_g1, which, row, rows);                        // A or B1
    const fr k = ld_fr(V.rs, which == 0 ? 1 : 0, row, rows);               // s for A, r for B1 (raw canonical words)
    glv_half h1, h2; fr_glv_split(k.v, h1, h2);
    if (part & 1u) P.X = fq_mul(P.X, fq_glv_beta());                       // phi in Jacobian coordinates: (beta X, Y, Z)
    const glv_half& h = (part & 1u) ? h2 : h1;
    st_g1_jac(tmp_g1, part, row, rows, jac_mul_u128_signed(P, h.mag, h.neg));
}
ZKP_HD inline void step_g16_final(const G16View& V, const uint32_t* sum_g1, const uint32_t* sum_g2, const uint32_t* tmp_g1, uint32_t which, uint32_t row) {
    const uint32_t rows = V.rows;
    uint8_t* o = V.out + (uint64_t)row * V.stride;
    uint32_t plen = 0; for (int k = 0; k < 4; k++) plen |= (uint32_t)o[2 + k] << (8 * k);
    uint8_t* pr = o + 10 + plen - 256;
    if (which == 0) {
        uint32_t w1[16]; g1_serialize(w1, ld_g1_jac(sum_g1, 0, row, rows)); g16_put_bytes(pr, w1, 16);
    } else if (which == 1) {
        uint32_t w2[32]; g2_serialize(w2, ld_g2_jac(sum_g2, 0, row, rows)); g16_put_bytes(pr + 64, w2, 32);
    } else {
        g1_jac Cc = ld_g1_jac(sum_g1, 2, row, rows);
        for (uint32_t t = 0; t < G16_CPARTS; t++) Cc = jac_add(Cc, ld_g1_jac(tmp_g1, t, row, rows));
        uint32_t w1[16]; g1_serialize(w1, Cc); g16_put_bytes(pr + 192, w1, 16);
    }
}

}  // namespace zkp
