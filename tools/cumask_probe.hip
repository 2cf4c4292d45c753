// Development probe: which XCDs / CUs do workgroups land on under a hipExtStreamCreateWithCUMask stream?
// (decides how the "narrow kernel" partition of the Bulletproofs pipeline must be masked)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <string>
__global__ void probe(uint32_t* out, int spin) {
    uint32_t xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    volatile uint32_t x = threadIdx.x;
    for (int i = 0; i < spin; i++) x = x * 1664525u + 1013904223u;
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid + (x & 0); }
}
static void run(const char* name, hipStream_t st, int nblocks) {
    uint32_t* d; hipMalloc(&d, 8 * nblocks);
    probe<<<nblocks, 64, 0, st>>>(d, 20000);
    hipStreamSynchronize(st);
    std::vector<uint32_t> h(2 * nblocks); hipMemcpy(h.data(), d, 8 * nblocks, hipMemcpyDeviceToHost);
    std::map<uint32_t, int> per_xcc; std::map<uint32_t, int> per_cu;
    for (int i = 0; i < nblocks; i++) { per_xcc[h[2 * i] & 0xf]++; per_cu[((h[2 * i] & 0xf) << 16) | ((h[2 * i + 1] >> 8) & 0xf) | (((h[2 * i + 1] >> 13) & 0x7) << 4) | (((h[2*i+1] >> 12) & 1) << 8)]++; }
    printf("%s: blocks %d, distinct (xcc,se,sh,cu) %zu; per xcc:", name, nblocks, per_cu.size());
    for (auto& kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
    printf("\n");
    hipFree(d);
}
// `--leak`: leave one CU-masked stream alive at exit (no hipStreamDestroy).  Round 1 saw an exit-time SIGSEGV inside
// __cxa_finalize under rocprofv3 in exactly the two runs whose builds created such streams and never destroyed them;
// this switch reproduces that condition in isolation ("exit-time teardown").
int main(int argc, char** argv) {
    const bool leak = argc > 1 && std::string(argv[1]) == "--leak";
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d\n", p.multiProcessorCount);
    hipStream_t s0; hipStreamCreate(&s0);
    run("unmasked", s0, 4096);
    for (int variant = 0; variant < 4; variant++) {
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const char* name = "";
        if (variant == 0) { mask[0] = 0x0000ffffu; name = "bits 0..15"; }
        if (variant == 1) { mask[0] = 0x01010101u; mask[1] = 0x01010101u; name = "bits 0,8,16,...,56"; }
        if (variant == 2) { for (int i = 0; i < 8; i++) mask[i] = 0xffffffffu; mask[0] = 0xffff0000u; name = "all but bits 0..15"; }
        if (variant == 3) { mask[0] = 0xffffffffu; name = "bits 0..31"; }
        hipStream_t s; hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
        if (e != hipSuccess) { printf("%s: create failed: %s\n", name, hipGetErrorString(e)); continue; }
        run(name, s, 4096);
        if (leak && variant == 3) { printf("leaving the CU-masked stream of variant 3 alive at exit\n"); continue; }
        hipStreamDestroy(s);
    }
    return 0;
}
