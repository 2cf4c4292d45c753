// Groth16 verification kernels of libzkp_hip (fifth translation unit).  An envelope's check is three data-dependent
// Miller loops and one final exponentiation (g16_verify.h, bn254_pairing.h): k_g16_pairs (lane = envelope) parses the
// points and accumulates the public inputs, k_g16_miller (lane = (envelope, pair)) runs the Miller loops three lanes per
// envelope (and B's subgroup check on a fourth), k_g16_finish (lane = envelope) multiplies them with the key's constant factor and exponentiates.  The code is the
// straightforward tower arithmetic with real (non-inlined) device functions, heavy on registers and scratch; a batch of
// 4096 envelopes is 64 / 192 / 64 waves, so the time is a lane's serial chain, which the split shortens.
#include <hip/hip_runtime.h>
#include "g16_verify.h"
using namespace zkp;

__global__ void __launch_bounds__(64) k_g16_pairs(int kind, const uint8_t* in, uint64_t stride, const uint32_t* len, uint32_t n, G16Vk vk, G16Pairs* pairs, uint8_t* valid) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint32_t l = len[i] <= stride ? len[i] : 0u;
    const uint8_t* env = in + (uint64_t)i * stride;
    G16Pairs o;
    const bool v = kind == G16_EQUALITY ? g16_equality_pairs(vk, env, l, o) : g16_membership_pairs(vk, env, l, o);
    valid[i] = v ? 1 : 0;
    if (v) pairs[i] = o;
}
// blockIdx.y = pair: the lanes of a wave share gamma (pair 1) or delta (pair 2); blockIdx.y = 3: B's subgroup check
// (a 254-bit scalar multiplication in G2, needed for the verdict only: it runs beside the Miller loops, not in front of them)
__global__ void __launch_bounds__(64) k_g16_miller(uint32_t n, const G16Pairs* pairs, const uint8_t* valid, fq12* f, uint8_t* sub_ok) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y;
    if (i >= n || !valid[i]) return;
    if (j == 3) sub_ok[i] = g16_b_in_subgroup(pairs[i]) ? 1 : 0;
    else f[(size_t)j * n + i] = g16_pair_miller(pairs[i], j);
}
__global__ void __launch_bounds__(64) k_g16_finish(uint32_t n, G16Vk vk, const uint8_t* valid, const uint8_t* sub_ok, const fq12* f, uint8_t* ok) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    ok[i] = valid[i] && sub_ok[i] && g16_finish(vk, f[i], f[(size_t)n + i], f[(size_t)2 * n + i]) ? 1 : 0;
}
size_t g16_verify_scratch_bytes(uint32_t n) { return (size_t)n * (sizeof(G16Pairs) + 3 * sizeof(fq12) + 2) + 256; }
void g16_launch_verify(int kind, const uint8_t* d_in, uint64_t stride, const uint32_t* d_len, uint32_t n, const G16Vk& vk, void* d_scratch, uint8_t* d_ok, hipStream_t st) {
    if (!n) return;
    fq12* f = reinterpret_cast<fq12*>(d_scratch);
    G16Pairs* pairs = reinterpret_cast<G16Pairs*>(f + (size_t)3 * n);
    uint8_t* valid = reinterpret_cast<uint8_t*>(pairs + n);
    uint8_t* sub_ok = valid + n;
    const uint32_t nb = (n + 63) / 64;
    k_g16_pairs<<<nb, 64, 0, st>>>(kind, d_in, stride, d_len, n, vk, pairs, valid);
    k_g16_miller<<<dim3(nb, 4), 64, 0, st>>>(n, pairs, valid, f, sub_ok);
    k_g16_finish<<<nb, 64, 0, st>>>(n, vk, valid, sub_ok, f, d_ok);
}
