#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int NACC>
__global__ void __launch_bounds__(256) k(uint64_t* out, uint32_t a, uint32_t b, int iters) {
    uint64_t acc[NACC];
    for (int k = 0; k < NACC; k++) acc[k] = threadIdx.x * 8 + k;
    uint32_t x = a + threadIdx.x, y = b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 128 / NACC; u++) {
#pragma unroll
            for (int k = 0; k < NACC; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(x), "v"(y) : "vcc");
        }
    }
    uint64_t s = 0; for (int k = 0; k < NACC; k++) s ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> void run(int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 4000;
    uint64_t* d; hipMalloc(&d, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 4; r++) { hipEventRecord(e0); k<NACC><<<blocks, 256>>>(d, 1u, 2u, iters); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    printf("chains/lane %d waves/SIMD %d : %.2f T mad/s\n", NACC, waves_per_simd, (double)blocks * 256 * iters * 128.0 / (best * 1e-3) / 1e12);
    hipFree(d);
}
int main() { for (int w : {1, 2, 3, 4, 8}) { run<1>(w); run<2>(w); run<4>(w); } return 0; }
