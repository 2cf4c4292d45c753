// Host-callable launcher of the Groth16 verification kernels (g16_verify_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include "g16_verify.h"

size_t g16_verify_scratch_bytes(uint32_t n);     // device scratch for n envelopes (Miller-loop values, parsed points)
void g16_launch_verify(int kind, const uint8_t* d_in, uint64_t stride, const uint32_t* d_len, uint32_t n, const zkp::G16Vk& vk, void* d_scratch, uint8_t* d_ok, hipStream_t st);
