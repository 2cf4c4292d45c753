// ed25519 fixed-base MSM over HBM-resident radix-2^16 tables (edg.h) and the kernels that build those tables: a translation unit of
// its own so that the Bulletproofs host code and the Groth16 kernels need not recompile with it.
#include "edg_launch.h"
#include "msm_kernel.h"

struct EdGather {      // edwards25519, affine-Niels entries gathered per lane from HBM, extended-coordinate accumulator
    static constexpr uint32_t ACC_W = GE_W, DIG_PER_WORD = 2;
    static __device__ __forceinline__ int32_t digit(uint32_t word, uint32_t w) { return (int32_t)(int16_t)(word >> (16 * (w & 1u))); }
    using Acc = ge;
    static __device__ __forceinline__ Acc identity() { return ge_identity(); }
#ifndef ZKP_EDG_WAVES
#define ZKP_EDG_WAVES 3
#define ZKP_EDG_PREFETCH 2
#endif
    // Three waves per SIMD with the entries of the next TWO steps in flight (2 x 24 registers, 155 VGPRs, no spills).  Measured (MI355X, 4096 x
    // prove_range staged, ms per batch): four waves and no lead 11.7 -- the 128-register budget leaves no room for an entry in flight and the
    // gather's latency shows on every step --, four waves with one step of lead (64 spilled registers) 11.5, four waves with the lead through
    // LDS by DMA 7.9, three waves with one step of lead 7.4, with two 7.5 (mixed batch: 12.6 / 12.4); the LDS-streamed radix-1024 kernel
    // this replaces: 9.6 (profiles/r04_edg_ab.jsonl)
    static constexpr uint32_t GATHER_WAVES = ZKP_EDG_WAVES; static constexpr int GATHER_PREFETCH = ZKP_EDG_PREFETCH;
    static constexpr uint32_t GATHER_W = EDG_ENTRY_W, GATHER_STRIDE = EDG_SLOT_W, GATHER_PRIO = 0;
    using GAcc = ge;
    static __device__ __forceinline__ GAcc to_gather(const ge& a) { return a; }
    static __device__ __forceinline__ ge from_gather(const GAcc& a) { return a; }
    static __device__ __forceinline__ GAcc accumulate_entry(const GAcc& acc, int32_t d, const uint32_t* e) { return edg_accumulate(acc, d, e); }
    static __device__ __forceinline__ void store(uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows, const Acc& a) { st_ge(p, idx, row, rows, a); }
    static __device__ __forceinline__ Acc load(const uint32_t* p, uint32_t idx, uint32_t row, uint32_t rows) { return ld_ge(p, idx, row, rows); }
    static __device__ __forceinline__ Acc add(const Acc& a, const Acc& b) { return ge_add(a, b); }
};
template __global__ void k_msm_gather<EdGather>(MsmView, uint32_t, uint32_t);
// The same kernel with its waves' issue priority raised (s_setprio; levels 1, 2 and 3 measured alike): for the chain's launches in a batch
// that also holds Groth16 work, whose G1 / G2 gather waves share the SIMDs (batch_impl.inc).
#ifndef ZKP_EDG_PRIO
#define ZKP_EDG_PRIO 1
#endif
struct EdGatherPrio : EdGather { static constexpr uint32_t GATHER_PRIO = ZKP_EDG_PRIO; };
template __global__ void k_msm_gather<EdGatherPrio>(MsmView, uint32_t, uint32_t);

uint32_t edg_msm_rows_per_block() { return 256; }
uint32_t edg_msm_blocks_per_cu() {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_msm_gather<EdGather>, 256, gather_lds_bytes<EdGather>()) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return occ > 0 ? (uint32_t)occ : 0;
}
void edg_launch_msm(const MsmView& m, uint32_t ngroups, uint32_t nblocks, hipStream_t st, bool raised) {
    const uint32_t grid = ((nblocks + 7) / 8) * 8;
    if (raised) k_msm_gather<EdGatherPrio><<<grid, 256, gather_lds_bytes<EdGather>(), st>>>(m, ngroups, nblocks);
    else k_msm_gather<EdGather><<<grid, 256, gather_lds_bytes<EdGather>(), st>>>(m, ngroups, nblocks);
}

// ---- table construction (edg.h: steps 1-5)
__global__ void __launch_bounds__(64) k_edg_bases(const uint32_t* gens, uint32_t* bases) {
    const uint32_t b = blockIdx.x * 64 + threadIdx.x;
    if (b < NBASE) edg_step_bases(gens, bases, b);
}
__global__ void __launch_bounds__(64) k_edg_starts(const uint32_t* bases, uint32_t* starts) {
    const uint32_t bw = blockIdx.x * 64 + threadIdx.x;
    if (bw < NBASE * EDG_NWIN) edg_step_starts(bases, starts, bw);
}
__global__ void __launch_bounds__(256) k_edg_fill(const uint32_t* bases, const uint32_t* starts, uint32_t* table) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;                 // (bw, run), run fastest
    if (t < NBASE * EDG_NWIN * EDG_NSEG) edg_step_fill(bases, starts, table, t / EDG_NSEG, t % EDG_NSEG);
}
__global__ void __launch_bounds__(256) k_edg_affine(uint32_t* table) {
    const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (g < (size_t)NBASE * EDG_NWIN * EDG_NENT / EDG_INV) edg_step_affine(table, g);
}
__global__ void __launch_bounds__(256) k_edg_check(const uint32_t* table, const uint32_t* gens, int* bad) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= (size_t)NBASE * EDG_NWIN * EDG_NENT) return;
    const uint32_t e = (uint32_t)(t % EDG_NENT), bw = (uint32_t)(t / EDG_NENT);
    if (!edg_step_check(table, gens, bw / EDG_NWIN, bw % EDG_NWIN, e)) atomicAdd(bad, 1);
}
size_t edg_build_scratch_words() { return (size_t)NBASE * EDG_NWIN * (EDG_NSEG + 1) * GE_W; }
void edg_launch_build(const uint32_t* d_gens, uint32_t* d_table, uint32_t* d_scratch, int* d_bad, hipStream_t st) {
    uint32_t* bases = d_scratch;                                        // [NBASE * EDG_NWIN][40]
    uint32_t* starts = d_scratch + (size_t)NBASE * EDG_NWIN * GE_W;     // [NBASE * EDG_NWIN][EDG_NSEG][40]
    k_edg_bases<<<(NBASE + 63) / 64, 64, 0, st>>>(d_gens, bases);
    k_edg_starts<<<(NBASE * EDG_NWIN + 63) / 64, 64, 0, st>>>(bases, starts);
    k_edg_fill<<<(NBASE * EDG_NWIN * EDG_NSEG + 255) / 256, 256, 0, st>>>(bases, starts, d_table);
    const size_t groups = (size_t)NBASE * EDG_NWIN * EDG_NENT / EDG_INV, slots = (size_t)NBASE * EDG_NWIN * EDG_NENT;
    k_edg_affine<<<(uint32_t)((groups + 255) / 256), 256, 0, st>>>(d_table);
    k_edg_check<<<(uint32_t)((slots + 255) / 256), 256, 0, st>>>(d_table, d_gens, d_bad);
}
