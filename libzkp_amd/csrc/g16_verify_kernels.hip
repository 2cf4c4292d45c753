// Groth16 verification kernel of libzkp_hip (fifth translation unit): lane = envelope, three Miller loops and one final
// exponentiation per proof (g16_verify.h, bn254_pairing.h).  Verification is not on the proving hot path: the code is
// the straightforward tower arithmetic with real (non-inlined) device functions, heavy on registers and scratch.
#include <hip/hip_runtime.h>
#include "g16_verify.h"
using namespace zkp;

__global__ void __launch_bounds__(64) k_g16_verify(int kind, const uint8_t* in, uint64_t stride, const uint32_t* len, uint32_t n, G16Vk vk, uint8_t* ok) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint32_t l = len[i] <= stride ? len[i] : 0u;
    const uint8_t* env = in + (uint64_t)i * stride;
    ok[i] = (kind == G16_EQUALITY ? g16_verify_equality_envelope(vk, env, l) : g16_verify_membership_envelope(vk, env, l)) ? 1 : 0;
}
void g16_launch_verify(int kind, const uint8_t* d_in, uint64_t stride, const uint32_t* d_len, uint32_t n, const G16Vk& vk, uint8_t* d_ok, hipStream_t st) {
    if (n) k_g16_verify<<<(n + 63) / 64, 64, 0, st>>>(kind, d_in, stride, d_len, n, vk, d_ok);
}
