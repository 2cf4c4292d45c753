// Common macros for the device math library.  Every arithmetic routine is `__host__ __device__` so
// that (a) the host side of the library can build generator tables with the very same code and
// (b) tests/emul can run the per-thread step functions on the CPU here (no GPU in the build container).
// The product path itself never computes proofs on the host: see zkp_hip.cpp.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZKP_HD __host__ __device__ __attribute__((always_inline))
#define ZKP_HD_NOINLINE __host__ __device__ __noinline__
#define ZKP_UNROLL _Pragma("unroll")
#else
#define ZKP_HD
#define ZKP_HD_NOINLINE
#define ZKP_UNROLL
#endif

// Register budget of the light chain kernels (k_poly, k_round_prep, k_round_sum, k_poly_sum).  In a mixed batch they start
// beside the Groth16 gather kernels, which hold three 136-VGPR waves on every SIMD: 104 of the 512 registers are left, so a kernel
// compiled for five waves per SIMD (<= 96 VGPRs) is resident the moment its launch is reached.  These four were at 80-106 and give
// up 2-15 spilled registers for it.  Tried in round 3 and NOT kept: the same cap on the heavy ones (k_encode 256 -> 96 VGPRs with 211
// spills, k_sum_t<EdMsm> 351, k_g16_cparts 478, k_g16_final 184, k_g16_witness 55): the mixed batch went from 13.7 to 14.5 ms and
// equality alone from 4.8 to 5.5 ms -- their time is instruction latency and the scratch round trips add to it; the transcript kernels
// cannot be capped at all (12.8 KB of LDS per 64-lane block bounds their occupancy at 3, and the compiler budgets registers for that).
#ifndef ZKP_LAT_WAVES
#define ZKP_LAT_WAVES 5
#endif

// Issue priority of a wave among the waves of its SIMD (s_setprio, 0..3; default 0).  The chain kernels -- a few waves each, their time is
// instruction latency -- raise it: in a mixed batch they share SIMDs with MSM waves that would otherwise take three of every four issue
// slots.  -DZKP_CHAIN_PRIO=0 builds without it (A/B).
#ifndef ZKP_CHAIN_PRIO
#define ZKP_CHAIN_PRIO 3
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define ZKP_RAISE_PRIO() __builtin_amdgcn_s_setprio(ZKP_CHAIN_PRIO)
#else
#define ZKP_RAISE_PRIO() ((void)0)
#endif
