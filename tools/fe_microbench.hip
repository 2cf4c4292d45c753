// Microbenchmark: which representation of GF(2^255-19) multiplies fastest on gfx950 VALU?
//   A: 8 x 32-bit saturated limbs, operand scanning (v_mad_u64_u32 + carry adds), fold by 38
//   B: 10 x 25.5-bit limbs, 64-bit column sums (v_mad_u64_u32), fold by 19
//   C: 12 x ~21.3-bit limbs held as position-scaled doubles (v_fma_f64), fold by 38*2^-256
// Each thread runs 4 interleaved dependent chains; prints Gmul/s and thread-0 results as integers
// (limb lists) so tools/check_fe_microbench.py can verify them with Python bigints.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

// ---------------------------------------------------------------- A: 8x32
struct FeA { uint32_t v[8]; };
__device__ __forceinline__ FeA mulA(const FeA& a, const FeA& b) {
    uint32_t r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint64_t t = (uint64_t)a.v[i] * b.v[j] + r[i + j] + carry;
            r[i + j] = (uint32_t)t;
            carry = (uint32_t)(t >> 32);
        }
        r[i + 8] = carry;
    }
    // fold high half: 2^256 = 38 mod p
    uint32_t carry = 0;
    FeA o;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t t = (uint64_t)r[i + 8] * 38u + r[i] + carry;
        o.v[i] = (uint32_t)t;
        carry = (uint32_t)(t >> 32);
    }
    // carry < 39: fold again (twice to be safe)
    uint64_t t = (uint64_t)o.v[0] + (uint64_t)carry * 38u;
    o.v[0] = (uint32_t)t; uint32_t c = (uint32_t)(t >> 32);
#pragma unroll
    for (int i = 1; i < 8; i++) { uint64_t u = (uint64_t)o.v[i] + c; o.v[i] = (uint32_t)u; c = (uint32_t)(u >> 32); }
    o.v[0] += c * 38u;  // cannot overflow again in practice (c=1 implies tiny low limbs)
    return o;
}

// ---------------------------------------------------------------- B: 10x25.5
struct FeB { uint32_t v[10]; };
__device__ __forceinline__ FeB mulB(const FeB& f, const FeB& g) {
    // limbs: even 26 bits, odd 25 bits. Inputs assumed < 2^26.x (after carry).
    uint32_t g19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) { g19[i] = 19u * g.v[i]; }
#pragma unroll
    for (int i = 0; i < 10; i++) { f2[i] = (i & 1) ? 2u * f.v[i] : f.v[i]; }
    uint64_t h[10];
#pragma unroll
    for (int k = 0; k < 10; k++) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            int j = k - i;
            if (j >= 0) {
                // odd*odd needs factor 2
                uint32_t fi = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
                acc += (uint64_t)fi * g.v[j];
            } else {
                j += 10;
                uint32_t fi = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
                acc += (uint64_t)fi * g19[j];
            }
        }
        h[k] = acc;
    }
    // carry chain
    FeB o;
    uint64_t c;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        int bits = (k & 1) ? 25 : 26;
        c = h[k] >> bits;
        h[k] &= ((1ull << bits) - 1);
        if (k < 9) h[k + 1] += c;
    }
    h[0] += c * 19;
    c = h[0] >> 26; h[0] &= ((1ull << 26) - 1); h[1] += c;
#pragma unroll
    for (int k = 0; k < 10; k++) o.v[k] = (uint32_t)h[k];
    return o;
}

// B1: the product form libzkp_amd/csrc/fe25519.h uses -- each column's v_mad_u64_u32 chain starts from the previous column's carry
__device__ __forceinline__ FeB mulB1(const FeB& f, const FeB& g) {
    uint32_t g19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) g19[i] = 19u * g.v[i];
#pragma unroll
    for (int i = 1; i < 10; i += 2) f2[i] = 2u * f.v[i];
    FeB o;
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        uint64_t acc = c;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            int j = k - i;
            const bool wrap = j < 0;
            if (wrap) j += 10;
            const uint32_t fi = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
            acc += (uint64_t)fi * (wrap ? g19[j] : g.v[j]);
        }
        const int bits = (k & 1) ? 25 : 26;
        o.v[k] = (uint32_t)acc & ((1u << bits) - 1);
        c = acc >> bits;
    }
    const uint64_t t = (uint64_t)o.v[0] + c * 19;
    o.v[0] = (uint32_t)t & 0x3ffffffu;
    o.v[1] += (uint32_t)(t >> 26);
    return o;
}
// B2: B1 with the 64-bit carry shift written as two 32-bit operations (funnel shift + shift)
__device__ __forceinline__ FeB mulB2(const FeB& f, const FeB& g) {
    uint32_t g19[10], f2[10];
#pragma unroll
    for (int i = 0; i < 10; i++) g19[i] = 19u * g.v[i];
#pragma unroll
    for (int i = 1; i < 10; i += 2) f2[i] = 2u * f.v[i];
    FeB o;
    uint64_t c = 0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        uint64_t acc = c;
#pragma unroll
        for (int i = 0; i < 10; i++) {
            int j = k - i;
            const bool wrap = j < 0;
            if (wrap) j += 10;
            const uint32_t fi = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
            acc += (uint64_t)fi * (wrap ? g19[j] : g.v[j]);
        }
        const int bits = (k & 1) ? 25 : 26;
        const uint32_t lo = (uint32_t)acc, hi = (uint32_t)(acc >> 32);
        o.v[k] = lo & ((1u << bits) - 1);
        c = ((uint64_t)(hi >> bits) << 32) | __builtin_amdgcn_alignbit(hi, lo, bits);
    }
    const uint64_t t = (uint64_t)o.v[0] + c * 19;
    o.v[0] = (uint32_t)t & 0x3ffffffu;
    o.v[1] += (uint32_t)(t >> 26);
    return o;
}

// ---------------------------------------------------------------- C: 12 doubles, position-scaled
// limb i holds a_i * 2^{p_i}, p_i = ceil(64*i/3): 0,22,43,64,86,107,128,150,171,192,214,235 ; p_12 = 256
struct FeC { double v[12]; };
__device__ __constant__ double kMagic[24];  // 1.5 * 2^(52 + p_{k+1}) for column k (k=0..22)
__device__ __forceinline__ FeC mulC(const FeC& a, const FeC& b) {
    double col[23];
#pragma unroll
    for (int k = 0; k < 23; k++) {
        double acc = 0.0;
        bool first = true;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            int j = k - i;
            if (j < 0 || j > 11) continue;
            if (first) { acc = a.v[i] * b.v[j]; first = false; }
            else acc = __builtin_fma(a.v[i], b.v[j], acc);
        }
        col[k] = acc;
    }
    const double fold = 38.0 * 0x1p-256;
#pragma unroll
    for (int k = 0; k < 11; k++) col[k] = __builtin_fma(col[k + 12], fold, col[k]);
    // carry propagate over columns 0..11 ; wrap to 0 with *38*2^-256
    FeC o;
    double carry = 0.0;
#pragma unroll
    for (int k = 0; k < 12; k++) {
        double x = col[k] + carry;
        double hi = (x + kMagic[k]) - kMagic[k];
        o.v[k] = x - hi;
        carry = hi;
    }
    // carry is a multiple of 2^256
    double x = __builtin_fma(carry, fold, o.v[0]);
    double hi = (x + kMagic[0]) - kMagic[0];
    o.v[0] = x - hi;
    o.v[1] += hi;
    return o;
}

template <typename Fe, Fe (*MUL)(const Fe&, const Fe&)>
__global__ void __launch_bounds__(256) bench_kernel(const Fe* in, Fe* out, int iters) {
    int tid = blockIdx.x * blockDim.x + threadIdx.x;
    Fe a = in[(tid * 4 + 0) % 64], b = in[(tid * 4 + 1) % 64], c = in[(tid * 4 + 2) % 64], d = in[(tid * 4 + 3) % 64];
    for (int it = 0; it < iters; it++) {
        a = MUL(a, b);
        b = MUL(b, c);
        c = MUL(c, d);
        d = MUL(d, a);
    }
    out[tid * 4 + 0] = a; out[tid * 4 + 1] = b; out[tid * 4 + 2] = c; out[tid * 4 + 3] = d;
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static const int P12[13] = {0, 22, 43, 64, 86, 107, 128, 150, 171, 192, 214, 235, 256};

template <typename Fe, Fe (*MUL)(const Fe&, const Fe&)>
static void run(const char* name, const std::vector<Fe>& hin, int blocks, int iters, void (*print)(const Fe&)) {
    Fe *din, *dout;
    size_t nthreads = (size_t)blocks * 256;
    CK(hipMalloc(&din, 64 * sizeof(Fe)));
    CK(hipMalloc(&dout, nthreads * 4 * sizeof(Fe)));
    CK(hipMemcpy(din, hin.data(), 64 * sizeof(Fe), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    bench_kernel<Fe, MUL><<<blocks, 256>>>(din, dout, 10);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    bench_kernel<Fe, MUL><<<blocks, 256>>>(din, dout, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double muls = (double)nthreads * 4.0 * iters;
    printf("{\"variant\": \"%s\", \"blocks\": %d, \"iters\": %d, \"ms\": %.3f, \"gmul_per_s\": %.2f}\n", name, blocks, iters, ms, muls / ms / 1e6);
    // correctness sample: small iteration count, thread 0
    bench_kernel<Fe, MUL><<<1, 256>>>(din, dout, 3);
    CK(hipDeviceSynchronize());
    std::vector<Fe> hout(4);
    CK(hipMemcpy(hout.data(), dout, 4 * sizeof(Fe), hipMemcpyDeviceToHost));
    printf("CHECK %s in", name);
    for (int i = 0; i < 4; i++) { printf(" "); print(hin[i]); }
    printf(" out");
    for (int i = 0; i < 4; i++) { printf(" "); print(hout[i]); }
    printf("\n");
    CK(hipFree(din)); CK(hipFree(dout));
}

static void printA(const FeA& x) { printf("A:"); for (int i = 0; i < 8; i++) printf("%u%s", x.v[i], i < 7 ? "," : ""); }
static void printB(const FeB& x) { printf("B:"); for (int i = 0; i < 10; i++) printf("%u%s", x.v[i], i < 9 ? "," : ""); }
static void printC(const FeC& x) { printf("C:"); for (int i = 0; i < 12; i++) printf("%.0f%s", ldexp(x.v[i], -P12[i]), i < 11 ? "," : ""); }

// ---------------------------------------------------------------- raw issue rate of v_mad_u64_u32
// 8 independent 64-bit accumulators per lane, 16 x 8 mads per loop trip, inline asm so nothing is folded away: anchors the
// field-multiplication ceilings (100 mads per GF(2^255-19) product, 200 per lazily reduced BN254 Fq product) to an
// instruction rate.  Also the same loop with full-rate v_add_u32 for the clock the chip holds on this kind of code.
__global__ void __launch_bounds__(256) k_mad_rate(uint64_t* out, uint32_t a, uint32_t b, int iters) {
    uint64_t acc[8];
    for (int k = 0; k < 8; k++) acc[k] = threadIdx.x * 8 + k;
    uint32_t x = a + threadIdx.x, y = b;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int k = 0; k < 8; k++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[k]) : "v"(x), "v"(y) : "vcc");
        }
    }
    uint64_t s = 0; for (int k = 0; k < 8; k++) s ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void __launch_bounds__(256) k_add_rate(uint32_t* out, uint32_t a, int iters) {
    uint32_t acc[8];
    for (int k = 0; k < 8; k++) acc[k] = threadIdx.x * 8 + k;
    uint32_t x = a + threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int k = 0; k < 8; k++) asm volatile("v_add_u32 %0, %1, %0" : "+v"(acc[k]) : "v"(x));
        }
    }
    uint32_t s = 0; for (int k = 0; k < 8; k++) s ^= acc[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static void run_rates(int blocks, int iters) {
    uint64_t* d; CK(hipMalloc(&d, (size_t)blocks * 256 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const double simds = prop.multiProcessorCount * 4.0;
    for (int which = 0; which < 2; which++) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; rep++) {
            CK(hipEventRecord(e0));
            if (which == 0) k_mad_rate<<<blocks, 256>>>(d, 12345u, 678901u, iters); else k_add_rate<<<blocks, 256>>>((uint32_t*)d, 12345u, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        const double ops = (double)blocks * 256 * iters * 128.0;
        const double rate = ops / (best * 1e-3);
        // wave-instructions per second per SIMD; at clock f a full-rate op issues one wave-instruction per 4 cycles
        const double per_simd = rate / 64.0 / simds;
        printf("{\"variant\": \"%s\", \"blocks\": %d, \"iters\": %d, \"ms\": %.3f, \"t_lane_ops_per_s\": %.3f, \"wave_instr_per_s_per_simd\": %.4g, \"cus\": %d, \"clock_mhz_reported\": %d}\n",
               which == 0 ? "v_mad_u64_u32_issue_rate" : "v_add_u32_issue_rate", blocks, iters, best, rate / 1e12, per_simd, prop.multiProcessorCount, prop.clockRate / 1000);
    }
    CK(hipFree(d));
}

int main(int argc, char** argv) {
    int blocks = argc > 1 ? atoi(argv[1]) : 256 * 8;
    int iters = argc > 2 ? atoi(argv[2]) : 2000;
    double magic[24];
    for (int k = 0; k < 23; k++) {
        int pnext = (k + 1 <= 12) ? P12[k + 1] : P12[k + 1 - 12] + 256;
        magic[k] = ldexp(1.5, 52 + pnext);
    }
    magic[23] = 0;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(kMagic), magic, sizeof(magic)));
    std::vector<FeA> ia(64); std::vector<FeB> ib(64); std::vector<FeC> ic(64);
    for (int n = 0; n < 64; n++) {
        for (int i = 0; i < 8; i++) ia[n].v[i] = (uint32_t)rnd();
        ia[n].v[7] &= 0x7fffffffu;
        for (int i = 0; i < 10; i++) ib[n].v[i] = (uint32_t)(rnd() & ((i & 1) ? 0x1ffffffu : 0x3ffffffu));
        for (int i = 0; i < 12; i++) { int w = P12[i + 1] - P12[i]; ic[n].v[i] = ldexp((double)(rnd() & ((1ull << w) - 1)), P12[i]); }
    }
    run_rates(blocks, iters);
    run<FeA, mulA>("A_8x32", ia, blocks, iters, printA);
    run<FeB, mulB>("B_10x25.5", ib, blocks, iters, printB);
    run<FeB, mulB1>("B1_10x25.5_seeded_carry", ib, blocks, iters, printB);
    run<FeB, mulB2>("B2_10x25.5_alignbit_carry", ib, blocks, iters, printB);
    run<FeC, mulC>("C_12xf64", ic, blocks, iters, printC);
    return 0;
}
