// Host-callable launchers of the Groth16 verification kernels (g16_verify_kernels.hip: one lane per chain; fq2vm_kernels.hip: the
// Fq2 virtual machine, K waves per chain).
#pragma once
#include <hip/hip_runtime.h>
#include <vector>
#include "g16_verify.h"

size_t g16_verify_scratch_bytes(uint32_t n);     // device scratch for n envelopes (Miller-loop values, parsed points)
void g16_launch_verify(int kind, const uint8_t* d_in, uint64_t stride, const uint32_t* d_len, uint32_t n, const zkp::G16Vk& vk, void* d_scratch, uint8_t* d_ok, hipStream_t st);

// ---- Fq2 virtual machine (fq2vm.h): micro-operation tables in device memory, one set per device
struct G16VmTables {
    const uint32_t *code = nullptr, *off = nullptr;                                 // four waves per chain (fq2vm_programs.h)
    const uint32_t* consts = nullptr;
    const uint16_t* script[4] = {nullptr, nullptr, nullptr, nullptr};               // chain A, subgroup, finish, chain B
    hipStream_t side[2] = {nullptr, nullptr}; hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};      // chain B and the subgroup chain run beside chain A
    hipStream_t vq[2] = {nullptr, nullptr};        // the batch check's virtual envelope: greatest priority, so that its one-workgroup chains get the first CU that drains
    hipStream_t sub = nullptr, tail = nullptr;      // the batch check of several rounds: its subgroup chain on every CU but the first of each XCD, the one-workgroup tail (product, finishing chain) on those eight
    bool ready = false;
};
int g16_vm_upload(G16VmTables& T);
void g16_vm_free(G16VmTables& T);
// per key: the machine's Miller value of (beta, -alpha) and the line table of gamma and delta (host computation through the same tables)
void g16_vm_key_constants(const zkp::g2_aff& beta, const zkp::g1_aff& neg_alpha, const zkp::g2_aff& gamma, const zkp::g2_aff& delta, uint32_t ml[6 * 20], std::vector<uint32_t>& lines);
size_t g16_vm_scratch_bytes(uint32_t n);
// verdicts of the generic envelopes in d_ok (0 for the others); *d_special (zeroed by the caller) counts the envelopes with a point at infinity in
// the proof, whose verdicts only g16_launch_verify gives
void g16_launch_verify_vm(int kind, const uint8_t* d_in, uint64_t stride, const uint32_t* d_len, uint32_t n, const zkp::G16Vk& vk, const G16VmTables& T, const uint32_t* d_kconst,
                          const uint32_t* d_lines, void* d_scratch, uint8_t* d_ok, uint32_t* d_special, hipStream_t st);
// One pairing check for the whole batch (g16_rlc.h).  d_scratch: g16_rlc_scratch_bytes(n, vk.n_ic) bytes with the weights (4 words per envelope,
// non-zero) at g16_rlc_rho_offset and four zeroed counters at g16_rlc_counters_offset: [0] envelopes with a point at infinity, [1] live envelopes
// whose B is outside the subgroup, [2] anomalies of the virtual envelope, [3] 1 when the batch's product is one.  d_ok holds the verdicts when
// counters == {0, 0, 0, 1}; otherwise the caller verifies envelope by envelope.
size_t g16_rlc_scratch_bytes(uint32_t n, uint32_t n_ic);
size_t g16_rlc_rho_offset(uint32_t n, uint32_t n_ic);
size_t g16_rlc_counters_offset(uint32_t n, uint32_t n_ic);
void g16_launch_verify_rlc(int kind, const uint8_t* d_in, uint64_t stride, const uint32_t* d_len, uint32_t n, const zkp::G16Vk& vk, const G16VmTables& T, const uint32_t* d_kconst,
                           const uint32_t* d_lines, void* d_scratch, uint8_t* d_ok, hipStream_t st);
