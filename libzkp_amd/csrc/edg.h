// Fixed-base tables of the 130 Bulletproofs generators sized for HBM, and the steps that build and read them.
//
// Round 4.  The ed25519 MSMs of the range prover (bulletproofs.rs:138,150 -> RangeProof::prove_single) walked
// radix-1024 tables streamed through LDS: 60 KB sub-tables, 26 additions per 253-bit scalar, one 1024-lane workgroup per CU.  Measured this
// round (tools/gather_calib.hip, profiles/r04_gather_calib.jsonl): lanes that gather ONE entry each from a sub-table of a few hundred KB
// somewhere in a multi-GB table are served at ~48 G gathers/s whether the entry is 64 or 128 bytes -- the memory system is bound by
// requests, not bytes, for this pattern -- so the tables move to HBM at radix 2^16: 16 windows of 32 768 affine-Niels entries per
// generator, 128-byte slots, 8.7 GB for the 130 generators; **16 additions per scalar instead of 26**, no LDS, no barrier, 256-lane
// workgroups of four independent waves that fit beside the Groth16 gather kernels' waves (a 1024-lane, 120 KB workgroup had to wait
// for a whole CU to drain).  Digits are limb-aligned: a 32-bit word of the scalar is two signed 16-bit digits.
//
// Entry e of (generator b, window w) = (e + 1) * 2^(16 w) * G_b as (y + x, y - x, 2 d x y), each the canonical 255-bit integer in
// eight 32-bit words; words 24..31 of the slot are padding (one 128-byte line per gather).
#pragma once
#include "bp_steps.h"

namespace zkp {

constexpr uint32_t EDG_WBITS = 16, EDG_NWIN = 16, EDG_NENT = 1u << 15, EDG_DIGW = 8, EDG_NWIN_U64 = 5;
constexpr uint32_t EDG_SLOT_W = 32, EDG_ENTRY_W = 24;                       // words per table slot / payload words
constexpr uint32_t EDG_SEG = 64, EDG_NSEG = EDG_NENT / EDG_SEG, EDG_INV = 8;   // builder: 64-entry runs, one inversion per 8 entries
ZKP_HD constexpr size_t edg_table_words() { return (size_t)NBASE * EDG_NWIN * EDG_NENT * EDG_SLOT_W; }
ZKP_HD constexpr size_t edg_slot(uint32_t b, uint32_t w, uint32_t e) { return (((size_t)b * EDG_NWIN + w) * EDG_NENT + e) * EDG_SLOT_W; }

ZKP_HD inline ge_niels edg_unpack(const uint32_t e[EDG_ENTRY_W]) {
    ge_niels n; n.ypx = fe_fromwords(e); n.ymx = fe_fromwords(e + 8); n.xy2d = fe_fromwords(e + 16);
    return n;
}
// acc + d * (window base), the entry |d| - 1 of the window already fetched
ZKP_HD inline ge edg_accumulate(const ge& acc, int32_t d, const uint32_t e[EDG_ENTRY_W]) {
    ge_niels n = edg_unpack(e);
    n = ge_niels_select(d < 0, ge_niels_neg(n), n);
    return ge_madd(acc, n);
}
// reference form of one chunk (host emulation, tests): entries read straight from the table
ZKP_HD inline ge edg_accumulate_from(const ge& acc, int32_t d, const uint32_t* table, uint32_t base, uint32_t w) {
    const uint32_t mag = (uint32_t)(d < 0 ? -d : d);
    return edg_accumulate(acc, d, table + edg_slot(base, w, mag - 1));
}

// ---- table construction (one-time, on the device; each function is the body of one lane).
// 1. window bases: 2^(16 w) * G_b for w = 0..15.            lane = generator
ZKP_HD inline void edg_step_bases(const uint32_t* gens /*[NBASE][40]*/, uint32_t* bases /*[NBASE][EDG_NWIN][40]*/, uint32_t b) {
    ge p = ld_ge(gens, b, 0, 1);
    for (uint32_t w = 0; w < EDG_NWIN; w++) {
        st_ge(bases, b * EDG_NWIN + w, 0, 1, p);
        for (uint32_t k = 0; k < EDG_WBITS; k++) p = ge_dbl(p);
    }
}
// 2. run starts: (64 s + 1) * P for s = 0..511, P = window base.     lane = (generator, window)
ZKP_HD inline void edg_step_starts(const uint32_t* bases, uint32_t* starts /*[NBASE * EDG_NWIN][EDG_NSEG][40]*/, uint32_t bw) {
    const ge p = ld_ge(bases, bw, 0, 1);
    ge step = p;
    for (uint32_t k = 0; k < 6; k++) step = ge_dbl(step);                 // 64 P
    ge acc = p;
    for (uint32_t s = 0; s < EDG_NSEG; s++) { st_ge(starts, (size_t)bw * EDG_NSEG + s, 0, 1, acc); acc = ge_add(acc, step); }
}
// 3. a run of 64 consecutive multiples, left in their slots as projective (X, Y, Z) limbs.     lane = (generator, window, run)
ZKP_HD inline void edg_step_fill(const uint32_t* bases, const uint32_t* starts, uint32_t* table, uint32_t bw, uint32_t s) {
    const ge p = ld_ge(bases, bw, 0, 1);
    ge acc = ld_ge(starts, (size_t)bw * EDG_NSEG + s, 0, 1);
    uint32_t* q = table + ((size_t)bw * EDG_NENT + (size_t)s * EDG_SEG) * EDG_SLOT_W;
    for (uint32_t e = 0; e < EDG_SEG; e++) {
        ZKP_UNROLL for (int k = 0; k < 10; k++) { q[k] = acc.X.v[k]; q[10 + k] = acc.Y.v[k]; q[20 + k] = acc.Z.v[k]; }
        q += EDG_SLOT_W;
        acc = ge_add(acc, p);
    }
}
// 4. eight slots from projective to packed affine Niels with ONE field inversion (Montgomery's trick).     lane = 8 consecutive slots
ZKP_HD inline fe edg_fe_invert(const fe& z) {              // z^(p-2) = (z^(2^252-3))^8 * z^3
    fe t = fe_pow22523(z);
    t = fe_sq(fe_sq(fe_sq(t)));
    return fe_mul(t, fe_mul(fe_sq(z), z));
}
ZKP_HD inline void edg_step_affine(uint32_t* table, size_t group) {
    uint32_t* q0 = table + group * EDG_INV * EDG_SLOT_W;
    fe pre[EDG_INV];
    fe run = fe_one();
    for (uint32_t i = 0; i < EDG_INV; i++) {
        fe z; ZKP_UNROLL for (int k = 0; k < 10; k++) z.v[k] = q0[(size_t)i * EDG_SLOT_W + 20 + k];
        pre[i] = run; run = fe_mul(run, z);
    }
    fe inv = edg_fe_invert(run);
    const fe d2 = fe_const_d2();
    for (uint32_t i = EDG_INV; i-- > 0;) {
        uint32_t* q = q0 + (size_t)i * EDG_SLOT_W;
        fe X, Y, Z; ZKP_UNROLL for (int k = 0; k < 10; k++) { X.v[k] = q[k]; Y.v[k] = q[10 + k]; Z.v[k] = q[20 + k]; }
        const fe zi = fe_mul(inv, pre[i]);
        inv = fe_mul(inv, Z);
        const fe x = fe_mul(X, zi), y = fe_mul(Y, zi);
        fe_towords(q, fe_add(y, x));
        fe_towords(q + 8, fe_sub(y, x));
        fe_towords(q + 16, fe_mul(fe_mul(x, y), d2));
        ZKP_UNROLL for (int k = EDG_ENTRY_W; k < (int)EDG_SLOT_W; k++) q[k] = 0;
    }
}
// 5. self-check of the finished table, all of it: entry[e] + entry[0] == entry[e + 1] inside a window, 2 * entry[32767] of window w ==
// entry[0] of window w + 1, and entry[0] of window 0 == the generator.  By induction every slot then holds the multiple it stands for.
// lane = (generator, window, e); returns false on a mismatch
ZKP_HD inline bool edg_niels_equals(const ge& p, const uint32_t e[EDG_ENTRY_W]) {       // projective p against an affine entry
    const ge_niels n = edg_unpack(e);
    return fe_eq(fe_add(p.Y, p.X), fe_mul(n.ypx, p.Z)) && fe_eq(fe_sub(p.Y, p.X), fe_mul(n.ymx, p.Z));
}
ZKP_HD inline ge edg_point_of(const uint32_t e[EDG_ENTRY_W]) {                          // affine entry -> extended point
    const ge_niels n = edg_unpack(e);
    // y = (ypx + ymx) / 2, x = (ypx - ymx) / 2: keep the factor 2 projectively (X : Y : Z) = (ypx - ymx : ypx + ymx : 2)
    ge p; p.X = fe_carry(fe_sub(n.ypx, n.ymx)); p.Y = fe_carry(fe_add(n.ypx, n.ymx)); p.Z = fe_zero(); p.Z.v[0] = 2;
    // T = X Y / Z: scale to (2X : 2Y : 4) so that T = X Y is exact
    p.T = fe_mul(p.X, p.Y); p.X = fe_carry(fe_add(p.X, p.X)); p.Y = fe_carry(fe_add(p.Y, p.Y)); p.Z.v[0] = 4;
    return p;
}
ZKP_HD inline bool edg_step_check(const uint32_t* table, const uint32_t* gens, uint32_t b, uint32_t w, uint32_t e) {
    const uint32_t* cur = table + edg_slot(b, w, e);
    if (e + 1 < EDG_NENT) {
        const ge sum = ge_madd(edg_point_of(cur), edg_unpack(table + edg_slot(b, w, 0)));
        if (!edg_niels_equals(sum, cur + EDG_SLOT_W)) return false;
    } else if (w + 1 < EDG_NWIN) {
        if (!edg_niels_equals(ge_dbl(edg_point_of(cur)), table + edg_slot(b, w + 1, 0))) return false;
    }
    if (w == 0 && e == 0 && !edg_niels_equals(ld_ge(gens, b, 0, 1), cur)) return false;
    return true;
}

}  // namespace zkp
