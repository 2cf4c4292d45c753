// Host-callable launcher of the Groth16 verification kernel (g16_verify_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "g16_verify.h"

void g16_launch_verify(int kind, const uint8_t* d_in, uint64_t stride, const uint32_t* d_len, uint32_t n, const zkp::G16Vk& vk, uint8_t* d_ok, hipStream_t st);
