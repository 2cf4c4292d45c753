// Host-callable launcher of the improvement-proof (STARK) kernel, defined in stark_kernels.hip (third translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include "stark_steps.h"

// one wavefront per proof; ops with new <= old produce out_len = 0 (the host front end reports them as InvalidInput)
void stark_launch_prove(const uint64_t* d_old, const uint64_t* d_new, uint32_t n, const zkp::StarkConst* d_const, uint8_t* d_out, uint64_t stride,
                        uint32_t* d_out_len, hipStream_t st);
// lane = envelope; ok[i] = 1 iff verify_improvement(proof_i, old_i) accepts
void stark_launch_verify(const uint8_t* d_in, uint64_t stride, const uint32_t* d_len, const uint64_t* d_old, uint32_t n, const zkp::StarkConst* d_const, uint8_t* d_ok,
                         hipStream_t st);
