// Host-callable launchers of the ed25519 gather MSM and of the builder of its HBM-resident tables (defined in edg_kernels.hip, a separate
// translation unit: the Bulletproofs host code in zkp_hip.hip calls these).
#pragma once
#include <hip/hip_runtime.h>
#include "edg.h"

uint32_t edg_msm_rows_per_block();            // lanes (= proofs) per MSM workgroup
uint32_t edg_msm_blocks_per_cu();             // resident workgroups per CU of the built kernel (occupancy query; 0 on failure)
void edg_launch_msm(const zkp::MsmView& m, uint32_t ngroups, uint32_t nblocks, hipStream_t st, bool raised = false);      // raised: the instantiation whose waves raise their issue priority
// builds the whole table on `st` from the 130 generators (extended coordinates, [NBASE][40] words on the device) and checks every slot
// against its neighbours; scratch: [NBASE * EDG_NWIN][EDG_NSEG + 1][40] words; *bad (device int, zeroed by the caller) counts mismatches
void edg_launch_build(const uint32_t* d_gens, uint32_t* d_table, uint32_t* d_scratch, int* d_bad, hipStream_t st);
size_t edg_build_scratch_words();
