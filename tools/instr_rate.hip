#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define KERNEL(name, decl, init, body)                                                             \
__global__ void __launch_bounds__(256) name(uint64_t* out, uint32_t a, uint32_t b, int iters) {   \
    decl; init;                                                                                    \
    uint32_t x = a + threadIdx.x, y = b; (void)x; (void)y;                                         \
    for (int it = 0; it < iters; it++) {                                                           \
        _Pragma("unroll") for (int u = 0; u < 16; u++) { _Pragma("unroll") for (int k = 0; k < 8; k++) { body; } } \
    }                                                                                              \
    uint64_t s = 0; for (int k = 0; k < 8; k++) s ^= (uint64_t)acc[k];                             \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                                \
}
KERNEL(k_mul_lo, uint32_t acc[8], for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k, asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(acc[k]) : "v"(y)))
KERNEL(k_mul_u24, uint32_t acc[8], for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k, asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(acc[k]) : "v"(y)))
KERNEL(k_lshr64, uint64_t acc[8], for (int k = 0; k < 8; k++) acc[k] = ((uint64_t)threadIdx.x << 40) + k, asm volatile("v_lshrrev_b64 %0, 1, %0" : "+v"(acc[k])))
KERNEL(k_lshladd64, uint64_t acc[8], for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k, asm volatile("v_lshl_add_u64 %0, %0, 0, %0" : "+v"(acc[k])))
KERNEL(k_and, uint32_t acc[8], for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k, asm volatile("v_and_b32 %0, %1, %0" : "+v"(acc[k]) : "v"(x)))
KERNEL(k_mad64, uint64_t acc[8], for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k, asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(x), "v"(y) : "vcc"))
KERNEL(k_mad_u32_u24, uint32_t acc[8], for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k, asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(x), "v"(y)))
KERNEL(k_alignbit, uint32_t acc[8], for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k, asm volatile("v_alignbit_b32 %0, %1, %0, 26" : "+v"(acc[k]) : "v"(x)))
KERNEL(k_mul_hi, uint32_t acc[8], for (int k = 0; k < 8; k++) acc[k] = threadIdx.x + k, asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(acc[k]) : "v"(y)))
template <class K> void run(const char* name, K kern) {
    const int blocks = 256 * 4, iters = 4000;
    uint64_t* d; (void)hipMalloc(&d, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 4; r++) { (void)hipEventRecord(e0); kern<<<blocks, 256>>>(d, 1u, 3u, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    printf("{\"instr\": \"%s\", \"t_lane_ops_per_s\": %.2f}\n", name, (double)blocks * 256 * iters * 128.0 / (best * 1e-3) / 1e12);
    (void)hipFree(d);
}
int main() {
    run("v_mad_u64_u32", k_mad64); run("v_mul_lo_u32", k_mul_lo); run("v_mul_hi_u32", k_mul_hi); run("v_mul_u32_u24", k_mul_u24); run("v_mad_u32_u24", k_mad_u32_u24);
    run("v_lshrrev_b64", k_lshr64); run("v_lshl_add_u64", k_lshladd64); run("v_and_b32", k_and); run("v_alignbit_b32", k_alignbit);
    return 0;
}
