// Range-proof verification kernels of libzkp_hip (steps: bp_verify.h).  The generator part of the check runs through the
// prover's fixed-base MSM, partial-sum and encode kernels (zkp_hip.hip); this unit holds the verifier-only steps.
#include "bpv_launch.h"
using namespace zkp;
static constexpr int VTW = 64, VTB = 256;

__global__ void __launch_bounds__(VTW) k_vparse(VfyView V, uint32_t n, uint64_t stride, const uint32_t* len, const uint64_t* mn, const uint64_t* mx) {
    const uint32_t i = blockIdx.x * VTW + threadIdx.x;
    if (i >= n) return;
    const uint32_t l = len[i] <= stride ? len[i] : 0u;          // a length beyond the stride is malformed input: reject
    step_vparse(V, i, V.in + (uint64_t)i * stride, (uint64_t)i * stride, l, mn[i], mx[i]);
}
__global__ void __launch_bounds__(VTW) k_vparse_threshold(VfyView V, uint32_t n, uint64_t stride, const uint32_t* len, const uint64_t* thr) {
    const uint32_t i = blockIdx.x * VTW + threadIdx.x;
    if (i >= n) return;
    const uint32_t l = len[i] <= stride ? len[i] : 0u;
    step_vparse_threshold(V, i, V.in + (uint64_t)i * stride, (uint64_t)i * stride, l, thr[i]);
}
__global__ void __launch_bounds__(VTW) k_vparse_consistency(VfyView V, uint32_t n, uint64_t stride, const uint32_t* len) {
    const uint32_t i = blockIdx.x * VTW + threadIdx.x;
    if (i >= n) return;
    const uint32_t l = len[i] <= stride ? len[i] : 0u;
    step_vparse_consistency(V, i, V.in + (uint64_t)i * stride, (uint64_t)i * stride, l);
}
__global__ void __launch_bounds__(VTB) k_vfinal_ranges(VfyView V, const uint32_t* enc, uint32_t n, uint8_t* ok) {
    const uint32_t i = blockIdx.x * VTB + threadIdx.x;
    if (i < n) step_vfinal_ranges(V, enc, i, ok);
}
__global__ void __launch_bounds__(VTW) k_vdecode(VfyView V) {
    const uint32_t job = blockIdx.x * VTW + threadIdx.x;
    if (job < V.M) step_vdecode(V, blockIdx.y, job);
}
// STROBE image of lane t at lds[i * VTW + t] (conflict-free), as in the prover's transcript kernels
__global__ void __launch_bounds__(VTW) k_vtranscript(VfyView V) {
    __shared__ uint32_t lds[50 * VTW];
    const uint32_t job = blockIdx.x * VTW + threadIdx.x;
    Strobe s; s.base = lds + threadIdx.x; s.stride = VTW; s.pos = 0; s.pos_begin = 0;
    if (job < V.M) step_vtranscript(V, job, s);
}
__global__ void __launch_bounds__(VTB) k_vscalars(VfyView V) {
    const uint32_t job = blockIdx.x * VTB + threadIdx.x;
    if (job < V.M) step_vscalars(V, blockIdx.y, job);
}
__global__ void __launch_bounds__(VTW) k_vvarbase(VfyView V) {
    const uint32_t job = blockIdx.x * VTW + threadIdx.x;
    if (job < V.M) step_vvarbase(V, blockIdx.y, job);
}
__global__ void __launch_bounds__(VTB) k_vfinal(VfyView V, const uint32_t* enc, uint32_t n, uint8_t* ok, uint32_t jobs_per) {
    const uint32_t i = blockIdx.x * VTB + threadIdx.x;
    if (i < n) step_vfinal(V, enc, i, ok, jobs_per);
}

void bpv_launch_parse(const VfyView& V, uint32_t n, uint64_t stride, const uint32_t* d_len, const uint64_t* d_min, const uint64_t* d_max, hipStream_t st) {
    k_vparse<<<(n + VTW - 1) / VTW, VTW, 0, st>>>(V, n, stride, d_len, d_min, d_max);
}
void bpv_launch_parse_threshold(const VfyView& V, uint32_t n, uint64_t stride, const uint32_t* d_len, const uint64_t* d_thr, hipStream_t st) {
    k_vparse_threshold<<<(n + VTW - 1) / VTW, VTW, 0, st>>>(V, n, stride, d_len, d_thr);
}
void bpv_launch_parse_consistency(const VfyView& V, uint32_t n, uint64_t stride, const uint32_t* d_len, hipStream_t st) {
    k_vparse_consistency<<<(n + VTW - 1) / VTW, VTW, 0, st>>>(V, n, stride, d_len);
}
void bpv_launch_final_ranges(const VfyView& V, const uint32_t* d_enc, uint32_t n, uint8_t* d_ok, hipStream_t st) { k_vfinal_ranges<<<(n + VTB - 1) / VTB, VTB, 0, st>>>(V, d_enc, n, d_ok); }
void bpv_launch_decode(const VfyView& V, hipStream_t st) { k_vdecode<<<dim3((V.M + VTW - 1) / VTW, VP_NUM), VTW, 0, st>>>(V); }
void bpv_launch_transcript(const VfyView& V, hipStream_t st) { k_vtranscript<<<(V.M + VTW - 1) / VTW, VTW, 0, st>>>(V); }
void bpv_launch_scalars(const VfyView& V, hipStream_t st) { k_vscalars<<<dim3((V.M + VTB - 1) / VTB, BP_N), VTB, 0, st>>>(V); }
void bpv_launch_varbase(const VfyView& V, hipStream_t st) { k_vvarbase<<<dim3((V.M + VTW - 1) / VTW, VP_NUM), VTW, 0, st>>>(V); }
void bpv_launch_final(const VfyView& V, const uint32_t* d_enc, uint32_t n, uint8_t* d_ok, uint32_t jobs_per, hipStream_t st) { k_vfinal<<<(n + VTB - 1) / VTB, VTB, 0, st>>>(V, d_enc, n, d_ok, jobs_per); }
