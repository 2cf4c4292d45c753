// Improvement-proof (Winterfell STARK) kernel of libzkp_hip: one 64-lane wavefront proves one (old, new) pair, its
// working set (two 64-row LDE columns, their BLAKE3 Merkle trees, the serialised proof) lives in ~14.5 KB of LDS.
// Algorithmic HBM traffic: 16 B in, <= 3527 B out per proof; the domain constants (4.4 KB) are shared by every proof.
#include "stark_launch.h"
using namespace zkp;

struct WaveSync { __device__ __forceinline__ void operator()() const { __syncthreads(); } };

__global__ void __launch_bounds__(64) k_stark_prove(const uint64_t* oldv, const uint64_t* newv, uint32_t n, const StarkConst* C, uint8_t* out, uint64_t stride, uint32_t* out_len) {
    __shared__ StarkMem M;
    const uint32_t row = blockIdx.x, tid = threadIdx.x;
    if (row >= n) return;
    const uint64_t o = oldv[row], w = newv[row];
    if (w <= o) { if (tid == 0) out_len[row] = 0; return; }          // uniform per block
    stark_prove(M, *C, o, w, tid, 64, WaveSync());
    const uint32_t len = M.out_len;
    uint8_t* dst = out + (uint64_t)row * stride;
    if (len <= stride) for (uint32_t i = tid; i < len; i += 64) dst[i] = M.out[i];
    if (tid == 0) out_len[row] = len <= stride ? len : 0;
}

// lane = envelope: a few hundred hash compressions and field products each, no shared state
__global__ void __launch_bounds__(64) k_stark_verify(const uint8_t* in, uint64_t stride, const uint32_t* len, const uint64_t* oldv, uint32_t n, const StarkConst* C, uint8_t* ok) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint32_t l = len[i] <= stride ? len[i] : 0u;
    ok[i] = stark_verify_envelope(in + (uint64_t)i * stride, l, oldv[i], *C) ? 1 : 0;
}
void stark_launch_verify(const uint8_t* d_in, uint64_t stride, const uint32_t* d_len, const uint64_t* d_old, uint32_t n, const StarkConst* d_const, uint8_t* d_ok, hipStream_t st) {
    if (n) k_stark_verify<<<(n + 63) / 64, 64, 0, st>>>(d_in, stride, d_len, d_old, n, d_const, d_ok);
}

void stark_launch_prove(const uint64_t* d_old, const uint64_t* d_new, uint32_t n, const StarkConst* d_const, uint8_t* d_out, uint64_t stride,
                        uint32_t* d_out_len, hipStream_t st) {
    if (n) k_stark_prove<<<n, 64, 0, st>>>(d_old, d_new, n, d_const, d_out, stride, d_out_len);
}
