// BN254 Fq Montgomery product on gfx950: the product's ten 26-bit limbs (bn254_fq.h: 100 + 100 multiply-adds) against nine 29-bit
// limbs (81 + 81; inputs must then be carried, there is no room for lazily added operands).  Four chained products per lane and
// iteration, as tools/fe_microbench.hip.   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ilibzkp_amd/csrc tools/fq_microbench.hip -o build/tools/fq_microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "bn254_fq.h"
using namespace zkp;

struct fq9 { uint32_t v[9]; };
__host__ __device__ constexpr uint32_t p9(int i) { constexpr uint32_t m[9] = {0x187cfd47u, 0x10460b6u, 0x1c72a34fu, 0x2d522d0u, 0x1585d978u, 0x2db40c0u, 0xa6e141u, 0xe5c2634u, 0x30644eu}; return m[i]; }
#define N0_29 0x4866389u
#define MASK29 0x1fffffffu
// a * b / 2^261 mod p; limbs < 2^29 in, carried out (value < a*b/2^261 + p)
__device__ __forceinline__ fq9 fq9_mul(const fq9& a, const fq9& b) {
    uint32_t m[9]; fq9 r; uint64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
#pragma unroll
        for (int j = 0; j <= i; j++) acc += (uint64_t)a.v[j] * b.v[i - j];
#pragma unroll
        for (int j = 0; j < i; j++) acc += (uint64_t)m[j] * p9(i - j);
        m[i] = ((uint32_t)acc * N0_29) & MASK29;
        acc += (uint64_t)m[i] * p9(0);
        acc >>= 29;
    }
#pragma unroll
    for (int i = 9; i < 17; i++) {
#pragma unroll
        for (int j = i - 8; j < 9; j++) acc += (uint64_t)a.v[j] * b.v[i - j];
#pragma unroll
        for (int j = i - 8; j < 9; j++) acc += (uint64_t)m[j] * p9(i - j);
        r.v[i - 9] = (uint32_t)acc & MASK29;
        acc >>= 29;
    }
    r.v[8] = (uint32_t)acc;
    return r;
}
__device__ __forceinline__ fq fq10_mul(const fq& a, const fq& b) { return fq_mul(a, b); }

template <typename Fe, Fe (*MUL)(const Fe&, const Fe&)>
__global__ void __launch_bounds__(256) bench_kernel(const Fe* in, Fe* out, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fe a = in[(tid * 4 + 0) % 64], b = in[(tid * 4 + 1) % 64], c = in[(tid * 4 + 2) % 64], d = in[(tid * 4 + 3) % 64];
    for (int it = 0; it < iters; it++) { a = MUL(a, b); b = MUL(b, c); c = MUL(c, d); d = MUL(d, a); }
    out[tid * 4 + 0] = a; out[tid * 4 + 1] = b; out[tid * 4 + 2] = c; out[tid * 4 + 3] = d;
}
template <typename Fe, Fe (*MUL)(const Fe&, const Fe&)>
static void run(const char* name, int nl, uint32_t mask, int blocks, int iters) {
    std::vector<Fe> h(64);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (auto& e : h) for (int i = 0; i < nl; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; e.v[i] = (uint32_t)s & (i == nl - 1 ? mask >> 8 : mask); }
    Fe *din, *dout; size_t nthreads = (size_t)blocks * 256;
    (void)hipMalloc(&din, 64 * sizeof(Fe)); (void)hipMalloc(&dout, nthreads * 4 * sizeof(Fe));
    (void)hipMemcpy(din, h.data(), 64 * sizeof(Fe), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    bench_kernel<Fe, MUL><<<blocks, 256>>>(din, dout, 10); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; r++) { (void)hipEventRecord(e0); bench_kernel<Fe, MUL><<<blocks, 256>>>(din, dout, iters); (void)hipEventRecord(e1); (void)hipDeviceSynchronize(); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    printf("{\"variant\": \"%s\", \"blocks\": %d, \"iters\": %d, \"ms\": %.3f, \"gmul_per_s\": %.2f}\n", name, blocks, iters, best, (double)nthreads * 4.0 * iters / best / 1e6);
    (void)hipFree(din); (void)hipFree(dout);
}
int main() {
    for (int blocks : {256 * 3, 256 * 8}) {
        run<fq, fq10_mul>("bn254_fq_10x26", 10, 0x3ffffffu, blocks, 2000);
        run<fq9, fq9_mul>("bn254_fq_9x29", 9, 0x1fffffffu, blocks, 2000);
    }
    return 0;
}
