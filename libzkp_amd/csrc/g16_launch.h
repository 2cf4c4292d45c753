// Host-callable launchers of the Groth16 kernels (defined in g16_kernels.hip, a separate translation unit so the two
// halves of the library compile in parallel).
#pragma once
#include <hip/hip_runtime.h>
#include "g16_steps.h"
#include "bp_steps.h"

void g16_launch_witness(const zkp::G16View& V, hipStream_t st);
void g16_launch_zdigits(const zkp::G16View& V, hipStream_t st);
hipError_t g16_launch_qap(const zkp::G16View& V, const zkp::G16Circuit& C, hipStream_t st);
void g16_launch_cparts(const zkp::G16View& V, const uint32_t* sum_g1, uint32_t* tmp_g1, hipStream_t st);
void g16_launch_final(const zkp::G16View& V, const uint32_t* sum_g1, const uint32_t* sum_g2, const uint32_t* tmp_g1, hipStream_t st);
void g16_launch_mimc(const uint64_t* values, uint32_t n, const uint32_t* mimc_c, uint8_t* out, hipStream_t st);
void g16_launch_build_table(bool g2, const uint32_t* bases, uint32_t nslots, uint32_t* table, hipStream_t st, bool msm_form, const zkp::G16Radix& rx);
uint32_t g16_table_entry_words(bool g2, bool msm_form);      // 32-bit words per stored table entry (packed 16 / 32 in the MSM form, 20 / 40 plain)
uint32_t g16_msm_rows_per_block(bool g2);      // lanes (= proofs) per MSM workgroup and resident workgroups per CU of the built kernel
uint32_t g16_msm_blocks_per_cu(bool g2);
void g16_launch_msm(bool g2, const zkp::MsmView& m, hipStream_t st);
void g16_launch_sum(bool g2, const zkp::ReduceView& R, uint32_t* sums, hipStream_t st);
void g16_launch_serialize(bool g2, const uint32_t* jac, uint32_t rows, uint8_t* out, hipStream_t st);
