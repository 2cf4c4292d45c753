// Point-type-generic fixed-base MSM and partial-sum kernels (shared by the Bulletproofs and Groth16 translation units).
#pragma once
#include <hip/hip_runtime.h>
#include "bp_steps.h"

using namespace zkp;
// digit radix and table shape are run-time properties of the launch (MsmView); the point arithmetic is a trait of the point type T
static constexpr int TW = 64;        // one wave per block for per-proof serial steps
static constexpr int TB = 256;       // threads per block for (i, proof) grids

// (Rounds 1-3 also had k_msm_dma here: radix-1024 sub-tables streamed through LDS by DMA for 1024-lane workgroups.  Round 4 moved the last of
// its users -- the ed25519 prover, then the verifier's fixed-base part -- to the gather kernel below and deleted it)

// ---- HBM-resident tables, per-lane gathers (Groth16 key points).  The tables are sized for HBM, not for LDS: radix 2^14
// needs 8192 entries per (key point, window) -- 512 KB for G1, 1 MB for G2, tens of GB per key in total; the shape (m.nwin, m.nent,
// m.digw) is a run-time property of the loaded key, smaller radices for GPUs with less free memory -- so each lane
// fetches the one entry its digit selects straight from global memory (the workgroups that share a chunk sit on one XCD and
// walk the same sub-tables at the same time, so a good part of the entries is served from that XCD's L2).  No LDS, no
// barrier: a workgroup is four independent waves.  Loads are software-pipelined when the point type has the registers for it
// (T::GATHER_PREFETCH = number of steps of lead, 0 / 1 / 2): an entry's 16-byte pieces are issued one or two additions (~10 us each)
// before the addition that consumes them; the digit word that selects them is fetched earlier still.  T::GATHER_PREFETCH = -1: one step
// of lead through LDS by DMA (no staging registers; gather_lds_bytes<T>() of dynamic LDS per workgroup).
template <class T> constexpr size_t gather_lds_bytes() { return T::GATHER_PREFETCH < 0 ? (size_t)4 * 2 * (T::GATHER_W / 4) * 64 * 16 : 0; }
// Round 4: the walk of a chunk is a FLAT STEP LIST read with scalar loads.  Until then every step began with a chain of dependent
// vector-memory loads of wave-uniform metadata (slot_nwin[s] -> wait -> slot_base[s] -> wait -> digit word -> wait -> the gather could be
// issued): the values are the same in all 64 lanes, but they came out of 8 / 16-bit arrays, which the compiler reads with
// global_load_ubyte / _ushort, and each of those queued behind the other waves' scattered gathers in the vector memory pipeline.
// MsmView::steps holds, for every step of every chunk, (index of the first table entry of the step's window, digit-word row << 1 | which
// half of the word): two dwords, fetched with s_load_dwordx2 from the scalar cache; chunk_step0[c] .. chunk_step0[c + 1] are chunk c's
// steps.  The per-lane digit words are loaded two steps before the gather they address is issued, so the only wait left in the loop is
// the one for the entry that the addition is about to consume, issued GATHER_PREFETCH steps earlier.
template <class T>
__global__ void __launch_bounds__(256, T::GATHER_WAVES) k_msm_gather(MsmView m, uint32_t ngroups, uint32_t nblocks) {
    constexpr uint32_t V4 = T::GATHER_W / 4;          // 16-byte pieces of one packed table entry (G1: 64 bytes, G2: 128, ed25519: 96)
    constexpr uint32_t SV4 = T::GATHER_STRIDE / 4;    // ... and of the slot it sits in (ed25519: 128-byte slots, one line per gather)
    const uint32_t tid = threadIdx.x;
    const uint32_t per_xcd = (nblocks + 7) / 8;
    const uint32_t linear = (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3);
    if (linear >= nblocks) return;
    const uint32_t chunk = linear / ngroups, group = linear % ngroups;
    const uint32_t row = group * 256u + tid;
    const bool active = row < m.rows;
    const uint32_t st0 = m.chunk_step0[chunk], left = m.chunk_step0[chunk + 1] - st0;          // wave-uniform: scalar loads
    const uint2* const steps = reinterpret_cast<const uint2*>(m.steps) + st0;
    typename T::GAcc acc = T::to_gather(m.acc_init ? T::load(m.acc_init, 0, 0, 1) : T::identity());    // the coordinate system of the loop
    const uint4* const table4 = reinterpret_cast<const uint4*>(m.table);
    auto digit_word = [&](const uint2 ds) -> uint32_t { return active ? m.digits[(size_t)(ds.y >> 1) * m.rows + row] : 0u; };
#if defined(ZKP_GATHER_EXPERIMENT) && ZKP_GATHER_EXPERIMENT == 1          // measurement builds only (tools/): every gather reads the table's first entry (always an L1 hit)
    auto entry_of = [&](const uint2, int32_t) -> const uint4* { return table4; };
#elif defined(ZKP_GATHER_EXPERIMENT) && ZKP_GATHER_EXPERIMENT == 2        // ... every lane of a wave reads the first entry of the step's window (one line per wave-step)
    auto entry_of = [&](const uint2 ds, int32_t) -> const uint4* { return table4 + (uint64_t)ds.x * SV4; };
#else
    auto entry_of = [&](const uint2 ds, int32_t d) -> const uint4* { return table4 + ((uint64_t)ds.x + (uint32_t)((d < 0 ? -d : d) - 1)) * SV4; };
#endif
    if (left == 0) { if (active) T::store(m.partial, chunk, row, m.rows, T::from_gather(acc)); return; }
    if constexpr (T::GATHER_PREFETCH < 0) {
#if defined(__HIP_DEVICE_COMPILE__)
        // Entries one step ahead THROUGH LDS (the G2 loop has no registers to spare: 241 VGPRs at two waves per SIMD).
        // global_load_lds_dwordx4 moves each lane's 16-byte pieces straight into LDS, no VGPR in between: piece k of the 64 lanes
        // of a wave lands as one contiguous KiB (lane l at + 16 l), two buffers per wave, 64 KB per workgroup for 128-byte entries.  A
        // wave only reads what it wrote itself: no barrier, its own vmcnt(0) is the hand-over.
        extern __shared__ uint4 gather_lds[];
        const uint32_t wave = tid >> 6, lane = tid & 63u;
        uint4* const wbuf = gather_lds + (size_t)wave * (2u * V4 * 64u);
        auto dma = [&](uint32_t buf, const uint2 ds, int32_t dd) {
            if (dd == 0) return;
            const uint4* src = entry_of(ds, dd);
            ZKP_UNROLL for (uint32_t k = 0; k < V4; k++) __builtin_amdgcn_global_load_lds(src + k, wbuf + (buf * V4 + k) * 64u, 16, 0, 0);
        };
        uint32_t cb = 0;
        int32_t d = T::digit(digit_word(steps[0]), steps[0].y & 1u);
        uint32_t q0 = left > 1 ? digit_word(steps[1]) : 0u, q1 = left > 2 ? digit_word(steps[2]) : 0u;      // digit words of steps i + 1, i + 2
        dma(0, steps[0], d);
        for (uint32_t i = 0; i < left; i++) {
            int32_t dn = 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this step's entry (issued a whole addition ago) and the digit words are there
            uint4 e[V4];
            if (d != 0) { ZKP_UNROLL for (uint32_t k = 0; k < V4; k++) e[k] = wbuf[(cb * V4 + k) * 64u + lane]; }
            if (i + 1 < left) { const uint2 ds = steps[i + 1]; dn = T::digit(q0, ds.y & 1u); dma(cb ^ 1u, ds, dn); }      // lands while this step's addition runs
            const uint32_t qn = i + 3 < left ? digit_word(steps[i + 3]) : 0u;
            if (d != 0) acc = T::accumulate_entry(acc, d, reinterpret_cast<const uint32_t*>(e));
            cb ^= 1u; d = dn; q0 = q1; q1 = qn;
        }
        if (active) T::store(m.partial, chunk, row, m.rows, T::from_gather(acc));
        return;
#endif
    } else {
        // Entries LEAD steps ahead in registers (LEAD = 0: fetched when the step begins).  e[0] / dd[0] belong to the step being added,
        // e[LEAD] is the one in flight; q0 / q1 are the digit words of the next two gathers to be issued.
        constexpr int LEAD = T::GATHER_PREFETCH;
        uint4 e[LEAD + 1][V4]; int32_t dd[LEAD + 1];
        auto fetch = [&](uint4* dst, const uint2 ds, int32_t d) {
            if (d == 0) return;
            const uint4* src = entry_of(ds, d);
            ZKP_UNROLL for (uint32_t k = 0; k < V4; k++) dst[k] = src[k];
        };
        uint32_t qpre[LEAD + 2];
        ZKP_UNROLL for (int k = 0; k < LEAD + 2; k++) qpre[k] = (uint32_t)k < left ? digit_word(steps[k]) : 0u;
        ZKP_UNROLL for (int k = 0; k < LEAD; k++) {
            dd[k] = 0;
            if ((uint32_t)k < left) { const uint2 ds = steps[k]; dd[k] = T::digit(qpre[k], ds.y & 1u); fetch(e[k], ds, dd[k]); }
        }
        uint32_t q0 = qpre[LEAD], q1 = qpre[LEAD + 1];
        if constexpr (T::GATHER_PRIO > 0) __builtin_amdgcn_s_setprio(T::GATHER_PRIO);          // issue priority over co-resident waves from here on
        for (uint32_t i = 0; i < left; i++) {
            dd[LEAD] = 0;
            if (i + LEAD < left) { const uint2 ds = steps[i + LEAD]; dd[LEAD] = T::digit(q0, ds.y & 1u); fetch(e[LEAD], ds, dd[LEAD]); }
            const uint32_t qn = i + LEAD + 2 < left ? digit_word(steps[i + LEAD + 2]) : 0u;
            if (dd[0] != 0) acc = T::accumulate_entry(acc, dd[0], reinterpret_cast<const uint32_t*>(e[0]));
            ZKP_UNROLL for (int k = 0; k < LEAD; k++) { ZKP_UNROLL for (uint32_t v = 0; v < V4; v++) e[k][v] = e[k + 1][v]; dd[k] = dd[k + 1]; }
            q0 = q1; q1 = qn;
        }
    }
    if (active) T::store(m.partial, chunk, row, m.rows, T::from_gather(acc));
}

// (Round 2 had tried a flat per-chunk step list once before, software-pipelined three deep so that an iteration only issues
// loads -- entry of step i+1, digit word of step i+2, descriptor of step i+3 -- and never waits behind one.  Kernel times were
// within 1 % of this loop (G1 4.01 vs 4.05 ms, G2 1.70 vs 1.67 ms at 1024 membership rows): the metadata loads at the head of an
// iteration are not what separates this kernel from the bare addition loop of tools/g1_add_rate.hip)
// Partial sums of one target.  A block of NTHREADS lanes owns ROWS consecutive rows and cuts the target's chunk partials into
// NTHREADS / ROWS slices per row: every lane adds its slice's partials (every (NTHREADS / ROWS)-th chunk), then a tree through LDS
// joins the slices.  The ed25519 sums (cheap additions) take 8 slices of 32 rows; the BN254 sums are chains of 15-20 Jacobian additions
// of ~30-60 us each on a handful of waves -- latency on the Groth16 prover's critical path -- and take 32 slices (8 additions deep).
// Blocks are 256 lanes: one wave per SIMD, so that in a mixed batch a block finds room beside the gather kernels' waves (a 512-lane
// block needed two free wave slots' worth of registers on all four SIMDs of a CU at once and waited milliseconds for them).
template <class T, uint32_t ROWS, uint32_t NTHREADS, uint32_t MIN_WAVES = 1>
__global__ void __launch_bounds__(NTHREADS, MIN_WAVES) k_sum_t(ReduceView R, uint32_t* sums) {
    ZKP_RAISE_PRIO();
    constexpr uint32_t SLICES = NTHREADS / ROWS;
    static_assert(NTHREADS % ROWS == 0 && (SLICES & (SLICES - 1)) == 0, "slices per row: a power of two");
    __shared__ uint32_t lds[T::ACC_W * (NTHREADS / 2)];              // [word][SLICES / 2 x ROWS]
    const uint32_t rl = threadIdx.x % ROWS, slice = threadIdx.x / ROWS;
    const uint32_t row = blockIdx.x * ROWS + rl, target = blockIdx.y;
    const bool active = row < R.rows;
    const uint32_t c0 = R.target_chunk_begin[target], c1 = R.target_chunk_begin[target + 1];
    typename T::Acc acc = T::identity();
    bool have = false;
    if (active) {
        for (uint32_t c = c0 + slice; c < c1; c += SLICES) {
            const typename T::Acc p = T::load(R.partial, c, row, R.rows);
            acc = have ? T::add(acc, p) : p;
            have = true;
        }
    }
    for (uint32_t stride = SLICES / 2; stride >= 1; stride >>= 1) {
        if (slice >= stride && slice < 2 * stride) T::store(lds, 0, (slice - stride) * ROWS + rl, NTHREADS / 2, acc);
        __syncthreads();
        if (slice < stride) acc = T::add(acc, T::load(lds, 0, slice * ROWS + rl, NTHREADS / 2));
        __syncthreads();
    }
    if (slice == 0 && active) {
        if (R.corr) acc = T::add(acc, T::load(R.corr, target, 0, 1));
        T::store(sums, target, row, R.rows, acc);
    }
}
