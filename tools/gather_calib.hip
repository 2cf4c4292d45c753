// Calibration of the HBM counters for the access pattern of k_msm_gather: every lane fetches ONE table entry from a random place of a table
// far larger than the Infinity Cache, as 16-byte loads (MI355X_MICROARCH.md: "other access widths are uncalibrated: calibrate on a known
// byte count in your own access pattern").  Modes:
//   0  64-byte entries on 64-byte boundaries, any half of a 128-byte line          (the G1 key tables of round 3)
//   1  128-byte entries on 128-byte boundaries                                     (the G2 key tables)
//   2  64-byte entries, always the FIRST half of a 128-byte line (the second half of every line is never wanted)
//   3  two 64-byte entries that share one 128-byte line (what pairing a_query[i] / b_g1_query[i] would fetch)
// Prints, per mode: gathers, bytes the lanes asked for, milliseconds, GB/s asked for.  Run it once plainly (rates) and once per counter set
// under `rocprofv3 --kernel-trace --pmc ...` (FETCH_SIZE; TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; TCC_REQ_sum TCC_MISS_sum TCC_HIT_sum):
// counter bytes per gather against 64 / 128 says whether a 64-byte gather is tallied -- and served -- as half a line or drags the whole line.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/gather_calib.hip -o build/tools/gather_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }

// region_lines != 0: the lanes of a workgroup draw from ONE contiguous region per iteration (a (key point, window) sub-table of the MSM:
// 8192 entries = 512 KB for G1), the region itself at a random place of the table; 0: every lane anywhere in the table.
template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const uint4* table, uint64_t lines, uint32_t iters, uint32_t* out, uint64_t region_lines) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x;
    uint4 acc = {0, 0, 0, 0};
    uint64_t h = mix(gid * 0x9e3779b97f4a7c15ull + 1), hb = mix(blockIdx.x * 0x2545f4914f6cdd1dull + 7);
    for (uint32_t it = 0; it < iters; it++) {
        h = mix(h + it);
        uint64_t line = h % lines;
        if (region_lines) { hb = mix(hb + it); line = (hb % (lines / region_lines)) * region_lines + h % region_lines; }
        const uint4* src = table + line * 8;                           // 8 x 16 bytes = one 128-byte line
        if (MODE == 0) src += ((h >> 40) & 1u) * 4;
        constexpr int PIECES = (MODE == 1 || MODE == 3) ? 8 : 4;
        uint4 e[PIECES];
#pragma unroll
        for (int k = 0; k < PIECES; k++) e[k] = src[k];
#pragma unroll
        for (int k = 0; k < PIECES; k++) { acc.x ^= e[k].x; acc.y += e[k].y; acc.z ^= e[k].z; acc.w += e[k].w; }
    }
    out[gid] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

int main(int argc, char** argv) {
    const uint64_t table_gb = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 8;
    const uint32_t iters = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 64;
    const uint32_t blocks = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 256 * 12;          // 3 waves per SIMD, as the G1 gather
    const uint64_t region_kb = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 0, region_lines = region_kb * 1024 / 128;
    const uint64_t table_mb = argc > 5 ? std::strtoull(argv[5], nullptr, 10) : 0;       // overrides table_gb (tables that fit the Infinity Cache)
    const uint64_t bytes = table_mb ? table_mb << 20 : table_gb << 30, lines = bytes / 128;
    uint4* table = nullptr; uint32_t* out = nullptr;
    CK(hipMalloc(&table, bytes)); CK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CK(hipMemset(table, 0x5a, bytes));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int mode = 0; mode < 4; mode++) {
        for (int rep = 0; rep < 2; rep++) {                            // rep 0 warms the TLBs
            CK(hipEventRecord(a));
            if (mode == 0) k_gather<0><<<blocks, 256>>>(table, lines, iters, out, region_lines);
            if (mode == 1) k_gather<1><<<blocks, 256>>>(table, lines, iters, out, region_lines);
            if (mode == 2) k_gather<2><<<blocks, 256>>>(table, lines, iters, out, region_lines);
            if (mode == 3) k_gather<3><<<blocks, 256>>>(table, lines, iters, out, region_lines);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
            if (rep == 1) {
                const double gathers = (double)blocks * 256 * iters, asked = gathers * ((mode == 1 || mode == 3) ? 128.0 : 64.0);
                std::printf("{\"mode\": %d, \"what\": \"%s\", \"table_MiB\": %llu, \"region_KiB\": %llu, \"blocks\": %u, \"gathers\": %.0f, \"bytes_asked\": %.0f, \"ms\": %.3f, \"asked_GBps\": %.1f, \"G_gathers_per_s\": %.2f}\n", mode,
                            mode == 0 ? "64 B entry, either half of a line" : mode == 1 ? "128 B entry = one line" : mode == 2 ? "64 B entry, first half of its line only" : "two 64 B entries sharing a line",
                            (unsigned long long)(bytes >> 20), (unsigned long long)region_kb, blocks, gathers, asked, ms, asked / ms / 1e6, gathers / ms / 1e6);
            }
        }
    }
    CK(hipFree(table)); CK(hipFree(out));
    return 0;
}
