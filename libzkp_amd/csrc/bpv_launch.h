// Host-callable launchers of the range-proof verification kernels (bpv_kernels.hip, fourth translation unit).
#pragma once
#include <hip/hip_runtime.h>
#include "bp_verify.h"

void bpv_launch_parse(const zkp::VfyView& V, uint32_t n, uint64_t stride, const uint32_t* d_len, const uint64_t* d_min, const uint64_t* d_max, hipStream_t st);
void bpv_launch_parse_threshold(const zkp::VfyView& V, uint32_t n, uint64_t stride, const uint32_t* d_len, const uint64_t* d_thr, hipStream_t st);
void bpv_launch_parse_consistency(const zkp::VfyView& V, uint32_t n, uint64_t stride, const uint32_t* d_len, hipStream_t st);
void bpv_launch_final_ranges(const zkp::VfyView& V, const uint32_t* d_enc, uint32_t n, uint8_t* d_ok, hipStream_t st);
void bpv_launch_decode(const zkp::VfyView& V, hipStream_t st);
void bpv_launch_transcript(const zkp::VfyView& V, hipStream_t st);
void bpv_launch_scalars(const zkp::VfyView& V, hipStream_t st);
void bpv_launch_varbase(const zkp::VfyView& V, hipStream_t st);
void bpv_launch_final(const zkp::VfyView& V, const uint32_t* d_enc, uint32_t n, uint8_t* d_ok, uint32_t jobs_per, hipStream_t st);
// batch check (random linear combination over all jobs, bucket-method MSM of the weighted proof points)
void bpv_launch_rlc_points(const zkp::VfyView& V, const zkp::RlcView& R, hipStream_t st);
void bpv_launch_rlc_sort(const zkp::RlcView& R, hipStream_t st);
void bpv_launch_rlc_reduce(const zkp::RlcView& R, hipStream_t st);
void bpv_launch_rlc_fixed_sum(const zkp::VfyView& V, uint32_t* d_digits1, hipStream_t st);
