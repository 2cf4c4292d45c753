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
