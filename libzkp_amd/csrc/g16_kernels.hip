// Groth16 / BN254 kernels of libzkp_hip (second translation unit; launchers declared in g16_launch.h).
#include "g16_launch.h"
#include "msm_kernel.h"

struct G1Msm {      // BN254 G1 key points (Groth16 a / b1 / h / l queries): packed affine table entries, XYZZ accumulator
    static constexpr uint32_t ACC_W = G1_JAC_W, DIG_PER_WORD = 2;
    static __device__ __forceinline__ int32_t digit(uint32_t word, uint32_t w) { return (int32_t)(int16_t)(word >> (16 * (w & 1u))); }
    using Acc = g1_jac;
    static __device__ __forceinline__ Acc identity() { return jac_infinity<fq>(); }
#ifndef ZKP_G1_GATHER_WAVES
#define ZKP_G1_GATHER_WAVES 3
#define ZKP_G1_GATHER_PREFETCH 2
#endif
    static constexpr uint32_t GATHER_WAVES = ZKP_G1_GATHER_WAVES; static constexpr int GATHER_PREFETCH = ZKP_G1_GATHER_PREFETCH;      // k_msm_gather: 3 waves/SIMD, entries two steps ahead
    // the gather loop accumulates in XYZZ coordinates on nine 29-bit limbs (bn254_g.h: g1_mmadd9); the key tables hold the
    // nine-limb coordinates packed into eight words each (fq9_pack8): a 64-byte, 64-byte-aligned entry x | y (k_g16_build_table)
    static constexpr uint32_t GATHER_W = 16, GATHER_STRIDE = 16, GATHER_PRIO = 0;
    using GAcc = g1_xyzz9;
    static __device__ __forceinline__ GAcc to_gather(const g1_jac& a) { return xyzz9_from_jac(a); }
    static __device__ __forceinline__ g1_jac from_gather(const GAcc& a) { return jac_from_xyzz9(a); }
    static __device__ __forceinline__ GAcc accumulate_entry(const GAcc& acc, int32_t d, const uint32_t* e) {
        g1_aff9 q{fq9_unpack8(e), fq9_unpack8(e + 8)};
        q.y = fq9_select(d < 0, fq9_neg_k<4>(q.y), q.y);                 // entries are < 3p
        return g1_mmadd9(acc, q);
    }
    static __device__ __forceinline__ void store(uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows, const Acc& a) { st_g1_jac(p, idx, row, rows, a); }
    static __device__ __forceinline__ Acc load(const uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows) { return ld_g1_jac(p, idx, row, rows); }
    static __device__ __forceinline__ Acc add(const Acc& a, const Acc& b) { return jac_add(a, b); }
};
struct G2Msm {      // BN254 G2 (Fq2 coordinates), Groth16 b_g2_query
    static constexpr uint32_t ACC_W = G2_JAC_W, DIG_PER_WORD = 2;
    static __device__ __forceinline__ int32_t digit(uint32_t word, uint32_t w) { return (int32_t)(int16_t)(word >> (16 * (w & 1u))); }
    using Acc = g2_jac;
    static __device__ __forceinline__ Acc identity() { return jac_infinity<fq2>(); }
#ifndef ZKP_G2_GATHER_PREFETCH
#define ZKP_G2_GATHER_PREFETCH (-1)
#endif
    static constexpr uint32_t GATHER_WAVES = 2; static constexpr int GATHER_PREFETCH = ZKP_G2_GATHER_PREFETCH;     // k_msm_gather: the addition itself takes ~240 VGPRs; entries one step ahead through LDS
    // the gather loop: XYZZ coordinates over Fq2 on nine 29-bit limbs (bn254_g.h: g2_mmadd9); table entries are the four
    // coordinates x.c0, x.c1, y.c0, y.c1 packed into eight words each: 128 bytes, one cache line
    static constexpr uint32_t GATHER_W = 32, GATHER_STRIDE = 32, GATHER_PRIO = 0;
    using GAcc = g2_xyzz9;
    static __device__ __forceinline__ GAcc to_gather(const g2_jac& a) { return g2_xyzz9_from_jac(a); }
    static __device__ __forceinline__ g2_jac from_gather(const GAcc& a) { return jac_from_g2_xyzz9(a); }
    static __device__ __forceinline__ GAcc accumulate_entry(const GAcc& acc, int32_t d, const uint32_t* e) {
        const g2_aff9 q{fq2_9{fq9_unpack8(e), fq9_unpack8(e + 8)}, fq2_9{fq9_unpack8(e + 16), fq9_unpack8(e + 24)}};
        return g2_mmadd9(acc, q, d < 0);
    }
    static __device__ __forceinline__ void store(uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows, const Acc& a) { st_g2_jac(p, idx, row, rows, a); }
    static __device__ __forceinline__ Acc load(const uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows) { return ld_g2_jac(p, idx, row, rows); }
    static __device__ __forceinline__ Acc add(const Acc& a, const Acc& b) { return jac_add(a, b); }
};


// ================================================================================================ kernels
__global__ void __launch_bounds__(TW) k_g16_witness(G16View V) {
    ZKP_RAISE_PRIO();
    const uint32_t row = blockIdx.x * TW + threadIdx.x;
    if (row < V.rows) step_g16_witness(V, row);
}
__global__ void __launch_bounds__(TB) k_g16_zdigits(G16View V) {
    ZKP_RAISE_PRIO();
    const uint32_t row = blockIdx.x * TB + threadIdx.x;
    if (row < V.rows) step_g16_zdigits(V, blockIdx.y, row);
}
struct DevSync { __device__ __forceinline__ void operator()() const { __syncthreads(); } };
// one workgroup of m / 4 lanes = one proof; ONE polynomial's nine limb rows in LDS at a time (g16_steps.h: g16_qap_proof)
static constexpr int QAP_TB_MAX = 256;
__global__ void __launch_bounds__(QAP_TB_MAX) k_g16_qap(G16View V, G16Circuit C) {
    extern __shared__ uint32_t g16_lds[];
    G16Lds L; L.base = g16_lds; L.m = C.m;
    g16_qap_proof(V, C, L, V.qap_evals + (size_t)blockIdx.x * 2 * 9 * C.m, blockIdx.x, threadIdx.x, blockDim.x, DevSync());
}
__global__ void __launch_bounds__(TW) k_g16_cparts(G16View V, const uint32_t* sum_g1, uint32_t* tmp_g1) {
    ZKP_RAISE_PRIO();
    const uint32_t row = blockIdx.x * TW + threadIdx.x;
    if (row < V.rows) step_g16_cparts(V, sum_g1, tmp_g1, blockIdx.y, row);
}
__global__ void __launch_bounds__(TW) k_g16_final(G16View V, const uint32_t* sum_g1, const uint32_t* sum_g2, const uint32_t* tmp_g1) {
    ZKP_RAISE_PRIO();
    const uint32_t row = blockIdx.x * TW + threadIdx.x;
    if (row < V.rows) step_g16_final(V, sum_g1, sum_g2, tmp_g1, blockIdx.y, row);
}
__global__ void __launch_bounds__(TW) k_mimc_commit(const uint64_t* values, uint32_t n, const uint32_t* mimc_c, uint8_t* out) {
    const uint32_t i = blockIdx.x * TW + threadIdx.x;
    if (i >= n) return;
    G16View V{}; V.mimc_c = mimc_c; V.z = nullptr;
    const fr h = g16_mimc_chain(V, 0, fp_from_u64<FrParams>(values[i]), 0);
    uint32_t w[8]; fp_to_raw(w, h);
    g16_put_bytes(out + 32ull * i, w, 8);
}
// Window tables of key points: entry e (1-based) of window w of point P is e * 2^(wbits w) * P in affine form.
// thread = (point, window, segment of SEG entries): the thread doubles P up to its window's power Q, steps to the first multiple of its
// segment ((seg SEG + 1) Q, a short double-and-add), then walks the segment by repeated Jacobian additions of Q, eight entries
// at a time with ONE field inversion per eight (Montgomery's trick).  Segments are what makes a key load short: with one thread per
// (point, window) -- round 2 -- a 8192-entry window was a serial chain of 8192 additions on a GPU holding only ~30 000 such threads
// (1.2 s per table, 2.4 s for both circuits); 512-entry segments give 16 times the threads and a sixteenth of the chain.
// fmt9: entries in the MSM loops' form (nine 29-bit limbs per Fq coordinate, < 2.4 p, packed into eight words: 64 / 128 bytes);
// otherwise plain ten-limb coordinates (the verifier's gamma_abc_g1 tables, g16_verify.h).
constexpr uint32_t G16_TABLE_SEG = 512;
template <class F, uint32_t AFF_W>
__global__ void __launch_bounds__(TW, 2) k_g16_build_table(const uint32_t* bases, uint32_t nslots, uint32_t* table, uint32_t fmt9, G16Radix rx) {
    const uint32_t seg_len = rx.nent < G16_TABLE_SEG ? rx.nent : G16_TABLE_SEG, nseg = rx.nent / seg_len, slot_seg = rx.slot_ent / seg_len;
    const uint32_t t = blockIdx.x * TW + threadIdx.x;
    if (t >= nslots * slot_seg) return;
    const uint32_t slot = t / slot_seg, sg = t % slot_seg;          // segment within the point's block -> (window, segment of the window)
    uint32_t win = sg / nseg, seg = sg % nseg;
    if (rx.uneven && sg >= 16u * nseg) { win = sg < 18u * nseg ? 16u : 17u; seg = sg - (win == 16u ? 16u : 18u) * nseg; }
    constexpr uint32_t FW = AFF_W / 2, BATCH = 8;
    Aff<F> base;
    {
        const uint32_t* b = bases + (size_t)slot * AFF_W;
        uint32_t* bx = reinterpret_cast<uint32_t*>(&base.x); uint32_t* by = reinterpret_cast<uint32_t*>(&base.y);
        for (uint32_t k = 0; k < FW; k++) { bx[k] = b[k]; by[k] = b[FW + k]; }
    }
    Jac<F> q = jac_from_aff(base);
    for (uint32_t i = 0; i < g16_win_bit(rx, win); i++) q = jac_dbl(q);
    // acc = (seg * seg_len + 1) q: binary ladder over the segment's first index (at most wbits - 1 bits)
    Jac<F> acc = q;
    {
        const uint32_t first = seg * seg_len + 1;
        int top = 31; while (top > 0 && !((first >> top) & 1u)) top--;
        for (int b = top - 1; b >= 0; b--) { acc = jac_dbl(acc); if ((first >> b) & 1u) acc = jac_add(acc, q); }
    }
    constexpr uint32_t NC = AFF_W / 10;                      // Fq coordinates per entry
    const uint32_t OW = fmt9 ? 8 * NC : AFF_W;             // words per stored entry
    uint32_t* dst = table + ((size_t)slot * rx.slot_ent + g16_win_off(rx, win) + (size_t)seg * seg_len) * OW;
    for (uint32_t e0 = 0; e0 < seg_len; e0 += BATCH) {
        Jac<F> pts[BATCH]; F zp[BATCH];
        for (uint32_t k = 0; k < BATCH; k++) {
            pts[k] = acc; acc = jac_add(acc, q);
            zp[k] = k ? f_mul(zp[k - 1], pts[k].Z) : pts[k].Z;
        }
        F inv = f_inv(zp[BATCH - 1]);
        for (int k = BATCH - 1; k >= 0; k--) {
            const F zi = k ? f_mul(inv, zp[k - 1]) : inv;
            if (k) inv = f_mul(inv, pts[k].Z);
            const F zi2 = f_sq(zi);
            Aff<F> a; a.x = f_mul(pts[k].X, zi2); a.y = f_mul(pts[k].Y, f_mul(zi2, zi));
            const uint32_t* ax = reinterpret_cast<const uint32_t*>(&a.x); const uint32_t* ay = reinterpret_cast<const uint32_t*>(&a.y);
            uint32_t* o = dst + (size_t)(e0 + k) * OW;
            if (fmt9) {                                            // the MSM loops' form: nine 29-bit limbs per Fq coordinate (< 2.4 p), packed
                const fq* c = reinterpret_cast<const fq*>(&a);     // x then y (G1) / x.c0, x.c1, y.c0, y.c1 (G2)
                for (uint32_t t2 = 0; t2 < NC; t2++) fq9_pack8(o + 8 * t2, fq9_from_fq(c[t2]));
                continue;
            }
            for (uint32_t j = 0; j < FW; j++) { o[j] = ax[j]; o[FW + j] = ay[j]; }
        }
    }
}



// thread = point: Jacobian sum -> ark-serialize uncompressed affine bytes (key generation output)
__global__ void __launch_bounds__(TW) k_g16_serialize(bool g2, const uint32_t* jac, uint32_t rows, uint8_t* out) {
    const uint32_t row = blockIdx.x * TW + threadIdx.x;
    if (row >= rows) return;
    if (!g2) { uint32_t w[16]; g1_serialize(w, ld_g1_jac(jac, 0, row, rows)); g16_put_bytes(out + 64ull * row, w, 16); }
    else { uint32_t w[32]; g2_serialize(w, ld_g2_jac(jac, 0, row, rows)); g16_put_bytes(out + 128ull * row, w, 32); }
}

template __global__ void k_msm_gather<G1Msm>(MsmView, uint32_t, uint32_t);
template __global__ void k_msm_gather<G2Msm>(MsmView, uint32_t, uint32_t);
static constexpr uint32_t G16_SUM_ROWS = 8, G16_SUM_TB = 256;         // 32 slices per row (msm_kernel.h)
template __global__ void k_sum_t<G1Msm, G16_SUM_ROWS, G16_SUM_TB>(ReduceView, uint32_t*);
template __global__ void k_sum_t<G2Msm, G16_SUM_ROWS, G16_SUM_TB>(ReduceView, uint32_t*);
template __global__ void k_g16_build_table<fq, 20>(const uint32_t*, uint32_t, uint32_t*, uint32_t, G16Radix);
template __global__ void k_g16_build_table<fq2, 40>(const uint32_t*, uint32_t, uint32_t*, uint32_t, G16Radix);

// ================================================================================================ launchers
void g16_launch_witness(const G16View& V, hipStream_t st) { k_g16_witness<<<(V.rows + TW - 1) / TW, TW, 0, st>>>(V); }
void g16_launch_zdigits(const G16View& V, hipStream_t st) { k_g16_zdigits<<<dim3((V.rows + TB - 1) / TB, V.nv), TB, 0, st>>>(V); }
hipError_t g16_launch_qap(const G16View& V, const G16Circuit& C, hipStream_t st) {
    const size_t lds = (size_t)9 * C.m * 4;             // one polynomial of nine-limb elements (18 KB at m = 512, 36 KB at m = 1024)
    const uint32_t tb = C.m / 4;
    if (tb < 64 || tb > (uint32_t)QAP_TB_MAX || (tb & 63u)) return hipErrorInvalidValue;      // domains of 256 .. 1024 points
    k_g16_qap<<<V.rows, tb, lds, st>>>(V, C);
    return hipSuccess;
}
void g16_launch_cparts(const G16View& V, const uint32_t* sum_g1, uint32_t* tmp_g1, hipStream_t st) {
    k_g16_cparts<<<dim3((V.rows + TW - 1) / TW, G16_CPARTS), TW, 0, st>>>(V, sum_g1, tmp_g1);
}
void g16_launch_final(const G16View& V, const uint32_t* sum_g1, const uint32_t* sum_g2, const uint32_t* tmp_g1, hipStream_t st) {
    k_g16_final<<<dim3((V.rows + TW - 1) / TW, 3), TW, 0, st>>>(V, sum_g1, sum_g2, tmp_g1);
}
void g16_launch_mimc(const uint64_t* values, uint32_t n, const uint32_t* mimc_c, uint8_t* out, hipStream_t st) { k_mimc_commit<<<(n + TW - 1) / TW, TW, 0, st>>>(values, n, mimc_c, out); }
// msm_form: the table feeds k_msm_gather (packed nine-limb entries); false = plain Fq limbs (the verifier's gamma_abc_g1 tables,
// g16_verify.h)
void g16_launch_build_table(bool g2, const uint32_t* bases, uint32_t nslots, uint32_t* table, hipStream_t st, bool msm_form, const G16Radix& rx) {
    const uint32_t seg_len = rx.nent < G16_TABLE_SEG ? rx.nent : G16_TABLE_SEG;
    const uint32_t threads = nslots * (rx.slot_ent / seg_len);
    if (!g2) k_g16_build_table<fq, 20><<<(threads + TW - 1) / TW, TW, 0, st>>>(bases, nslots, table, msm_form ? 1u : 0u, rx);
    else k_g16_build_table<fq2, 40><<<(threads + TW - 1) / TW, TW, 0, st>>>(bases, nslots, table, msm_form ? 1u : 0u, rx);
}
uint32_t g16_table_entry_words(bool g2, bool msm_form) { return msm_form ? (g2 ? G2Msm::GATHER_W : G1Msm::GATHER_W) : (g2 ? 40u : 20u); }
uint32_t g16_msm_rows_per_block(bool) { return 256u; }
uint32_t g16_msm_blocks_per_cu(bool g2) { return g2 ? G2Msm::GATHER_WAVES : G1Msm::GATHER_WAVES; }
void g16_launch_msm(bool g2, const MsmView& m, hipStream_t st) {      // HBM-resident tables, per-lane gathers; m.nwin / nent / digw = the key's radix
    const uint32_t ngroups = (m.rows + 255u) / 256u, nblocks = m.nchunks * ngroups, grid = ((nblocks + 7) / 8) * 8;
    if constexpr (gather_lds_bytes<G2Msm>() > 0 || gather_lds_bytes<G1Msm>() > 0) {      // opt in to the dynamic LDS of the DMA-prefetch form (every call: the attribute is per device)
        if (gather_lds_bytes<G1Msm>()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_gather<G1Msm>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gather_lds_bytes<G1Msm>());
        if (gather_lds_bytes<G2Msm>()) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_msm_gather<G2Msm>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gather_lds_bytes<G2Msm>());
    }
    if (!g2) k_msm_gather<G1Msm><<<grid, 256, gather_lds_bytes<G1Msm>(), st>>>(m, ngroups, nblocks);
    else k_msm_gather<G2Msm><<<grid, 256, gather_lds_bytes<G2Msm>(), st>>>(m, ngroups, nblocks);
}
void g16_launch_sum(bool g2, const ReduceView& R, uint32_t* sums, hipStream_t st) {
    const dim3 grid((R.rows + G16_SUM_ROWS - 1) / G16_SUM_ROWS, R.ntargets);
    if (!g2) k_sum_t<G1Msm, G16_SUM_ROWS, G16_SUM_TB><<<grid, G16_SUM_TB, 0, st>>>(R, sums);
    else k_sum_t<G2Msm, G16_SUM_ROWS, G16_SUM_TB><<<grid, G16_SUM_TB, 0, st>>>(R, sums);
}
void g16_launch_serialize(bool g2, const uint32_t* jac, uint32_t rows, uint8_t* out, hipStream_t st) { k_g16_serialize<<<(rows + TW - 1) / TW, TW, 0, st>>>(g2, jac, rows, out); }
