// How v_mad_u64_u32 shares the issue slots of a SIMD with simple VALU work, per waves/SIMD: per 8 multiply-adds a lane also
// issues NADD independent v_add_u32 (0, 4, 8, 16).  If the additions ride in the shadow of the multiply-adds the time per
// iteration stays that of the pure multiply-add stream.   hipcc -O3 --offload-arch=gfx950 tools/mad_mix.hip -o build/tools/mad_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int NADD>
__global__ void __launch_bounds__(256) k(uint64_t* out, uint32_t a, uint32_t b, int iters) {
    uint64_t acc[4];
    uint32_t s[4];
    for (int k = 0; k < 4; k++) { acc[k] = threadIdx.x * 8 + k; s[k] = threadIdx.x + k; }
    uint32_t x = a + threadIdx.x, y = b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {            // 16 groups of 8 multiply-adds + NADD additions
#pragma unroll
            for (int k = 0; k < 8; k++) {
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k & 3]) : "v"(x), "v"(y) : "vcc");
                if (NADD >= 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(s[k & 3]) : "v"(x));
                if (NADD >= 16) asm volatile("v_add_u32 %0, %0, %1" : "+v"(s[(k + 1) & 3]) : "v"(y));
                if (NADD == 4 && (k & 1)) asm volatile("v_add_u32 %0, %0, %1" : "+v"(s[k & 3]) : "v"(x));
            }
        }
    }
    uint64_t r = 0; for (int k = 0; k < 4; k++) r ^= acc[k] + s[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NADD> void run(int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 2000;
    uint64_t* d; hipMalloc(&d, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 4; r++) { hipEventRecord(e0); k<NADD><<<blocks, 256>>>(d, 1u, 2u, iters); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    const double mads = (double)blocks * 256 * iters * 128.0;
    printf("{\"adds_per_8_mads\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"t_mad_per_s\": %.2f, \"t_valu_per_s\": %.2f}\n", NADD, waves_per_simd, best, mads / (best * 1e-3) / 1e12, mads * (8 + NADD) / 8.0 / (best * 1e-3) / 1e12);
    hipFree(d);
}
int main() { for (int w : {1, 2, 3, 4, 6, 8}) { run<0>(w); run<4>(w); run<8>(w); run<16>(w); } return 0; }
