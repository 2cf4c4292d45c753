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

// ================================================================================================ batch check
// One random linear combination over the whole batch (upstream's verify_batch idea; weights from OS randomness per call):
//   sum_j rho_j [ generator part_j + sum_p s_jp P_jp ] == 0.
// The generator parts collapse to ONE 130-term row (k_rlc_fixed_sum -> the prover's fixed-base MSM with a single row); the 17 M
// proof-point terms are a variable-base MSM, the one place of this library where the bucket method pays: signed radix-2048
// digits, a counting sort per window (histogram by atomics, scan, scatter), one lane per bucket for the accumulation, then a
// segmented running sum and a Horner pass over the windows.
__device__ __forceinline__ void st_ge_flat(uint32_t* p, size_t idx, const ge& g) {
    uint32_t* q = p + idx * GE_W;
    ZKP_UNROLL for (int k = 0; k < 10; k++) { q[k] = g.X.v[k]; q[10 + k] = g.Y.v[k]; q[20 + k] = g.Z.v[k]; q[30 + k] = g.T.v[k]; }
}
__device__ __forceinline__ ge ld_ge_flat(const uint32_t* p, size_t idx) {
    ge g; const uint32_t* q = p + idx * GE_W;
    ZKP_UNROLL for (int k = 0; k < 10; k++) { g.X.v[k] = q[k]; g.Y.v[k] = q[10 + k]; g.Z.v[k] = q[20 + k]; g.T.v[k] = q[30 + k]; }
    return g;
}
// thread = (p, job): Niels form of the decoded point, digits of its weighted scalar, bucket histogram
__global__ void __launch_bounds__(VTB) k_rlc_points(VfyView V, RlcView R) {
    const uint32_t job = blockIdx.x * VTB + threadIdx.x, p = blockIdx.y;
    if (job >= V.M) return;
    const uint32_t idx = p * V.M + job;
    const bool live = !V.bad[job] && vpoint_present(p, V.lgn[job]);
    uint32_t pk[(RLC_NWIN + 1) / 2];
    ZKP_UNROLL for (uint32_t k = 0; k < (RLC_NWIN + 1) / 2; k++) pk[k] = 0;
    if (live) {
        const ge pt = ld_ge(V.pts, p, job, V.M);                     // Z = 1 (ge_ristretto_decode)
        uint32_t w8[8];
        fe_towords(w8, fe_add(pt.Y, pt.X)); const fe ypx = fe_fromwords(w8);
        fe_towords(w8, fe_sub(pt.Y, pt.X)); const fe ymx = fe_fromwords(w8);
        fe_towords(w8, fe_mul(pt.T, fe_const_d2())); const fe xy2d = fe_fromwords(w8);
        uint32_t* q = R.niels + (size_t)idx * RLC_NIELS_W;
        ZKP_UNROLL for (int k = 0; k < 10; k++) { q[k] = ypx.v[k]; q[10 + k] = ymx.v[k]; q[20 + k] = xy2d.v[k]; }
        sc_recode_signed<(int)RLC_WBITS, (int)RLC_NWIN>(pk, ld_sc(V.vscal, p, job, V.M));
    }
    for (uint32_t w = 0; w < RLC_NWIN; w++) {
        const int32_t d = (int32_t)(int16_t)(pk[w >> 1] >> (16 * (w & 1u)));
        R.dig[(size_t)w * R.N + idx] = (int16_t)d;
        if (d != 0) atomicAdd(&R.count[w * (RLC_NBUCKET + 1) + (uint32_t)(d < 0 ? -d : d)], 1u);
    }
}
// block = window: exclusive prefix sum of the NBUCKET + 1 bucket counts (entry 0, magnitude zero, is never used)
__global__ void __launch_bounds__(RLC_NBUCKET) k_rlc_scan(RlcView R) {
    __shared__ uint32_t part[RLC_NBUCKET];
    const uint32_t w = blockIdx.x, t = threadIdx.x;
    const uint32_t* c = R.count + w * (RLC_NBUCKET + 1);
    const uint32_t a = c[t + 1];                                     // magnitude t + 1
    part[t] = a;
    __syncthreads();
    for (uint32_t d = 1; d < RLC_NBUCKET; d <<= 1) { const uint32_t v = t >= d ? part[t - d] : 0; __syncthreads(); part[t] += v; __syncthreads(); }
    uint32_t* s = R.start + w * (RLC_NBUCKET + 2); uint32_t* cur = R.cursor + w * (RLC_NBUCKET + 1);
    s[t + 1] = part[t] - a; cur[t + 1] = part[t] - a;
    if (t == RLC_NBUCKET - 1) s[RLC_NBUCKET + 1] = part[t];
    if (t == 0) { s[0] = 0; cur[0] = 0; }
}
// thread = (point, window): place the point in its bucket's run
__global__ void __launch_bounds__(VTB) k_rlc_scatter(RlcView R) {
    const uint32_t idx = blockIdx.x * VTB + threadIdx.x, w = blockIdx.y;
    if (idx >= R.N) return;
    const int32_t d = R.dig[(size_t)w * R.N + idx];
    if (d == 0) return;
    const uint32_t pos = atomicAdd(&R.cursor[w * (RLC_NBUCKET + 1) + (uint32_t)(d < 0 ? -d : d)], 1u);
    R.sorted[(size_t)w * R.N + pos] = idx | (d < 0 ? 0x80000000u : 0u);
}
// four lanes per (window, bucket): each sums every fourth point of the bucket's run, two shuffle steps join them
__device__ __forceinline__ ge ge_shfl_xor(const ge& g, int mask) {
    ge r;
    ZKP_UNROLL for (int k = 0; k < 10; k++) {
        r.X.v[k] = __shfl_xor(g.X.v[k], mask); r.Y.v[k] = __shfl_xor(g.Y.v[k], mask); r.Z.v[k] = __shfl_xor(g.Z.v[k], mask); r.T.v[k] = __shfl_xor(g.T.v[k], mask);
    }
    return r;
}
__global__ void __launch_bounds__(VTW) k_rlc_buckets(RlcView R) {
    const uint32_t t = (blockIdx.x * VTW + threadIdx.x) >> 2, part = threadIdx.x & 3u;      // grid is exact: NWIN * NBUCKET * 4 lanes
    const uint32_t w = t / RLC_NBUCKET, b = t % RLC_NBUCKET + 1;
    const uint32_t* s = R.start + w * (RLC_NBUCKET + 2);
    ge acc = ge_identity();
    for (uint32_t k = s[b] + part; k < s[b + 1]; k += 4) {
        const uint32_t e = R.sorted[(size_t)w * R.N + k];
        acc = msm_accumulate_digit(acc, (e >> 31) ? -1 : 1, R.niels + (size_t)(e & 0x7fffffffu) * RLC_NIELS_W);
    }
    acc = ge_add(acc, ge_shfl_xor(acc, 1));
    acc = ge_add(acc, ge_shfl_xor(acc, 2));
    if (part == 0) st_ge_flat(R.bucket, t, acc);
}
// thread = (window, segment of 32 buckets): sum_b b * B_b over the segment = running sums + (segment base) * (segment total)
__global__ void __launch_bounds__(VTW) k_rlc_segments(RlcView R) {
    const uint32_t t = blockIdx.x * VTW + threadIdx.x;
    if (t >= RLC_NWIN * RLC_NSEG) return;
    const uint32_t w = t / RLC_NSEG, sgm = t % RLC_NSEG, base = sgm * RLC_SEG;
    ge run = ge_identity(), acc = ge_identity();
    for (int k = (int)RLC_SEG - 1; k >= 0; k--) {
        run = ge_add(run, ld_ge_flat(R.bucket, (size_t)w * RLC_NBUCKET + base + k));
        acc = ge_add(acc, run);                                      // acc = sum (k + 1) * B[base + k]
    }
    ge sh = ge_identity();                                           // base * run (base < NBUCKET, MSB first)
    for (int bit = (int)RLC_WBITS - 2; bit >= 0; bit--) { sh = ge_dbl(sh); if ((base >> bit) & 1u) sh = ge_add(sh, run); }
    st_ge_flat(R.seg, t, ge_add(acc, sh));
}
// one wave: lane w sums its window's segments; lane 0 runs Horner over the windows, adds the generator part, encodes
// One doubling split over four lanes of a wave: every lane holds p, lane role r squares one of X, Y, Z, X+Y and then forms one of
// the four closing products; v_readlane hands the results round.  A lone wave issues a dependent instruction chain at a fraction
// of the SIMD's rate, so the 253 doublings of the closing Horner chain are latency, and this cuts it four ways.
__device__ __forceinline__ fe fe_readlane(const fe& x, int lane) {
    fe r; ZKP_UNROLL for (int k = 0; k < 10; k++) r.v[k] = (uint32_t)__builtin_amdgcn_readlane((int)x.v[k], lane);
    return r;
}
__device__ __forceinline__ ge ge_dbl_lanes(const ge& p, uint32_t role) {
    const fe xy = fe_add(p.X, p.Y);
    const fe in = fe_select(role == 0, p.X, fe_select(role == 1, p.Y, fe_select(role == 2, p.Z, xy)));
    const fe s = fe_sq(in);
    const fe A = fe_readlane(s, 0), B = fe_readlane(s, 1), zz = fe_readlane(s, 2), t = fe_readlane(s, 3);
    const fe C = fe_add(zz, zz);
    const fe Hn = fe_add(A, B), En = fe_sub(Hn, t), Gn = fe_sub(A, B), Fn = fe_carry(fe_add(C, Gn));      // as ge_dbl
    const fe a = fe_select(role == 1, Gn, fe_select(role == 2, Fn, En));                                  // En*Fn, Gn*Hn, Fn*Gn, En*Hn
    const fe b = fe_select(role == 0, Fn, fe_select(role == 2, Gn, Hn));
    const fe m = fe_mul(a, b);
    ge r; r.X = fe_readlane(m, 0); r.Y = fe_readlane(m, 1); r.Z = fe_readlane(m, 2); r.T = fe_readlane(m, 3);
    return r;
}
// one block: eight lanes per window join its segment sums, then wave 0 runs the Horner chain over the windows
constexpr uint32_t RLC_FINAL_T = 256;
static_assert(RLC_NWIN * 8 <= RLC_FINAL_T && RLC_NSEG % 8 == 0, "eight lanes per window");
__global__ void __launch_bounds__(RLC_FINAL_T) k_rlc_final(RlcView R) {
    __shared__ uint32_t lds[RLC_NWIN * GE_W];
    const uint32_t t = threadIdx.x, w = t >> 3, q = t & 7u;
    if (t < RLC_NWIN * 8) {                                   // whole groups of eight within a wave: the shuffles stay inside the group
        ge acc = ld_ge_flat(R.seg, (size_t)w * RLC_NSEG + q);
        for (uint32_t k = q + 8; k < RLC_NSEG; k += 8) acc = ge_add(acc, ld_ge_flat(R.seg, (size_t)w * RLC_NSEG + k));
        acc = ge_add(acc, ge_shfl_xor(acc, 1));
        acc = ge_add(acc, ge_shfl_xor(acc, 2));
        acc = ge_add(acc, ge_shfl_xor(acc, 4));
        if (q == 0) st_ge_flat(lds, w, acc);
    }
    __syncthreads();
    if (t >= 64) return;
    const uint32_t role = t & 3u;
    ge acc = ld_ge_flat(lds, RLC_NWIN - 1);
    for (int k = (int)RLC_NWIN - 2; k >= 0; k--) {
        for (uint32_t d = 0; d < RLC_WBITS; d++) acc = ge_dbl_lanes(acc, role);
        acc = ge_add(acc, ld_ge_flat(lds, k));
    }
    if (t != 0) return;
    acc = ge_add(acc, ld_ge_flat(R.fixed_sum, 0));
    uint32_t e[8]; ge_ristretto_encode(e, acc);
    uint32_t o = 0; for (int k = 0; k < 8; k++) { R.result[k] = e[k]; o |= e[k]; }
    R.result[8] = o == 0 ? 1u : 0u;
}
// block = generator: sum of its weighted coefficients over the jobs -> digit row of the single-row fixed-base MSM
__global__ void __launch_bounds__(VTB) k_rlc_fixed_sum(VfyView V, uint32_t* digits1) {
    __shared__ uint32_t lds[8 * VTB];
    const uint32_t g = blockIdx.x, t = threadIdx.x;
    sc acc = sc_zero();
    for (uint32_t j = t; j < V.M; j += VTB) acc = sc_add(acc, ld_sc(V.fterm, g, j, V.M));
    for (uint32_t stride = VTB / 2; stride >= 1; stride >>= 1) {
        if (t >= stride && t < 2 * stride) { ZKP_UNROLL for (int k = 0; k < 8; k++) lds[k * VTB + t] = acc.v[k]; }
        __syncthreads();
        if (t < stride) { sc o; ZKP_UNROLL for (int k = 0; k < 8; k++) o.v[k] = lds[k * VTB + t + stride]; acc = sc_add(acc, o); }
        __syncthreads();
    }
    if (t == 0) st_digits(digits1, g, 0, 1, acc, V.dig16);
}
void bpv_launch_rlc_points(const VfyView& V, const RlcView& R, hipStream_t st) { k_rlc_points<<<dim3((V.M + VTB - 1) / VTB, VP_NUM), VTB, 0, st>>>(V, R); }
void bpv_launch_rlc_sort(const RlcView& R, hipStream_t st) {
    k_rlc_scan<<<RLC_NWIN, RLC_NBUCKET, 0, st>>>(R);
    k_rlc_scatter<<<dim3((R.N + VTB - 1) / VTB, RLC_NWIN), VTB, 0, st>>>(R);
}
void bpv_launch_rlc_reduce(const RlcView& R, hipStream_t st) {
    k_rlc_buckets<<<RLC_NWIN * RLC_NBUCKET * 4 / VTW, VTW, 0, st>>>(R);
    k_rlc_segments<<<(RLC_NWIN * RLC_NSEG + VTW - 1) / VTW, VTW, 0, st>>>(R);
    k_rlc_final<<<1, RLC_FINAL_T, 0, st>>>(R);
}
void bpv_launch_rlc_fixed_sum(const VfyView& V, uint32_t* d_digits1, hipStream_t st) { k_rlc_fixed_sum<<<NBASE, VTB, 0, st>>>(V, d_digits1); }

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
