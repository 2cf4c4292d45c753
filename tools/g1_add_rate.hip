// The ALU ceiling of the Groth16 G1 MSM loop: lanes add a point of a small in-register set to an XYZZ accumulator over and
// over (g1_mmadd_lazy, the loop body of k_msm_gather<G1Msm>, with no table gathers, no digits, no metadata), at 1..4 waves per
// SIMD.  The MSM kernel's additions/s against this figure is what its memory side and loop control cost; this figure
// against 1 810 multiply-adds at the v_mad_u64_u32 issue rate is what the field arithmetic's non-multiply instructions cost.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ilibzkp_amd/csrc tools/g1_add_rate.hip -o build/tools/g1_add_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include "bn254_g.h"
using namespace zkp;

template <int MAXW>
__global__ void __launch_bounds__(256, MAXW) k9(const uint32_t* pts, uint32_t* out, int iters, int per_lane) {      // the nine-limb form (g1_mmadd9)
    g1_aff9 q[2];
    // per_lane: every lane works on its own operands (256 different point pairs), as the MSM's lanes do; otherwise all lanes hold the
    // same limbs, which toggles far fewer bits in the multipliers -- the two cases differ in sustained clock on a power-limited chip
    const uint32_t* src = pts + (per_lane ? 40u * threadIdx.x : 0u);
    for (int j = 0; j < 2; j++) for (int k = 0; k < 9; k++) { q[j].x.v[k] = src[j * 20 + k]; q[j].y.v[k] = src[j * 20 + 10 + k]; }
    g1_xyzz9 acc; acc.X = q[0].x; acc.Y = q[0].y; acc.ZZ = q[1].x; acc.ZZZ = q[1].y;
    acc.X.v[0] ^= threadIdx.x & 1u;
    for (int it = 0; it < iters; it++) {
        g1_aff9 e = q[it & 1];
        e.y = fq9_select((it >> 1) & 1, fq9_neg_k<4>(e.y), e.y);
        acc = g1_mmadd9(acc, e);
    }
    uint32_t s = 0; for (int k = 0; k < 9; k++) s ^= acc.X.v[k] ^ acc.Y.v[k] ^ acc.ZZ.v[k] ^ acc.ZZZ.v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// ---- round 4, an alternative column form: two of the five carry passes of the addition folded away.  The nine-limb
// form has no room for limb-wise lazy sums in general (a column of 9 + 9 products of 29-bit limbs leaves under two bits of the 64-bit
// accumulator), but ONE operand of a product may be loose: Y3 = R (Q - X3) + (4p - Y1) PPP takes (Q - X3 + 8p) and (4p - Y1) limb by limb
// from "fat" forms of 8p / 4p (every limb below the top in [2^29 - 1, 2^30), same integer), no carry pass: limbs < 2^30.6, column sums
// 9 x 2^59.6 + 9 x 2^59 + 9 x 2^58 < 2^63.7.  Saves 2 x 18 of the ~2 070 issue units of an addition.
// (adopted: this is g1_mmadd9 of bn254_g.h since round 4; the carried form is kept below for the comparison)
__device__ __forceinline__ g1_xyzz9 g1_mmadd9_loose(const g1_xyzz9& p, const g1_aff9& q) {
    const fq9 U2 = fq9_mul(q.x, p.ZZ), S2 = fq9_mul(q.y, p.ZZZ);
    const fq9 P = fq9_sub_k<8>(U2, p.X), Rv = fq9_sub_k<4>(S2, p.Y);
    const fq9 PP = fq9_sq(P);
    const fq9 PPP = fq9_mul(P, PP), Q = fq9_mul(p.X, PP);
    const fq9 RR = fq9_sq(Rv);
    g1_xyzz9 r;
    r.X = fq9_sub2_k4(RR, PPP, Q);
    r.Y = fq9_mul_add2(Rv, fq9_sub_loose<8>(Q, r.X), fq9_neg_loose<4>(p.Y), PPP);
    r.ZZ = fq9_mul(p.ZZ, PP);
    r.ZZZ = fq9_mul(p.ZZZ, PPP);
    return r;
}
__device__ __forceinline__ g1_xyzz9 g1_mmadd9_carried(const g1_xyzz9& p, const g1_aff9& q) {       // rounds 2-3: every difference carried
    const fq9 U2 = fq9_mul(q.x, p.ZZ), S2 = fq9_mul(q.y, p.ZZZ);
    const fq9 P = fq9_sub_k<8>(U2, p.X), Rv = fq9_sub_k<4>(S2, p.Y);
    const fq9 PP = fq9_sq(P);
    const fq9 PPP = fq9_mul(P, PP), Q = fq9_mul(p.X, PP);
    const fq9 RR = fq9_sq(Rv);
    g1_xyzz9 r;
    r.X = fq9_sub2_k4(RR, PPP, Q);
    r.Y = fq9_mul_add2(Rv, fq9_sub_k<8>(Q, r.X), fq9_neg_k<4>(p.Y), PPP);
    r.ZZ = fq9_mul(p.ZZ, PP);
    r.ZZZ = fq9_mul(p.ZZZ, PPP);
    return r;
}
template <int MAXW>
__global__ void __launch_bounds__(256, MAXW) k9c(const uint32_t* pts, uint32_t* out, int iters, int per_lane) {
    g1_aff9 q[2];
    const uint32_t* src = pts + (per_lane ? 40u * threadIdx.x : 0u);
    for (int j = 0; j < 2; j++) for (int k = 0; k < 9; k++) { q[j].x.v[k] = src[j * 20 + k]; q[j].y.v[k] = src[j * 20 + 10 + k]; }
    g1_xyzz9 acc; acc.X = q[0].x; acc.Y = q[0].y; acc.ZZ = q[1].x; acc.ZZZ = q[1].y;
    acc.X.v[0] ^= threadIdx.x & 1u;
    for (int it = 0; it < iters; it++) {
        g1_aff9 e = q[it & 1];
        e.y = fq9_select((it >> 1) & 1, fq9_neg_k<4>(e.y), e.y);
        acc = g1_mmadd9_carried(acc, e);
    }
    uint32_t s = 0; for (int k = 0; k < 9; k++) s ^= acc.X.v[k] ^ acc.Y.v[k] ^ acc.ZZ.v[k] ^ acc.ZZZ.v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MAXW>
__global__ void __launch_bounds__(256, MAXW) k9l(const uint32_t* pts, uint32_t* out, int iters, int per_lane) {
    g1_aff9 q[2];
    const uint32_t* src = pts + (per_lane ? 40u * threadIdx.x : 0u);
    for (int j = 0; j < 2; j++) for (int k = 0; k < 9; k++) { q[j].x.v[k] = src[j * 20 + k]; q[j].y.v[k] = src[j * 20 + 10 + k]; }
    g1_xyzz9 acc; acc.X = q[0].x; acc.Y = q[0].y; acc.ZZ = q[1].x; acc.ZZZ = q[1].y;
    acc.X.v[0] ^= threadIdx.x & 1u;
    for (int it = 0; it < iters; it++) {
        g1_aff9 e = q[it & 1];
        e.y = fq9_select((it >> 1) & 1, fq9_neg_k<4>(e.y), e.y);
        acc = g1_mmadd9_loose(acc, e);
    }
    uint32_t s = 0; for (int k = 0; k < 9; k++) s ^= acc.X.v[k] ^ acc.Y.v[k] ^ acc.ZZ.v[k] ^ acc.ZZZ.v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static void run_loose(int waves, const uint32_t* d_pts) {
    const int blocks = 256 * waves, iters = 400;
    uint32_t* d; (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best[2] = {1e30f, 1e30f};
    for (int r = 0; r < 6; r++) for (int v = 0; v < 2; v++) {                     // alternately, so that both forms see the same clocks
        (void)hipEventRecord(e0);
        if (v == 0) k9c<3><<<blocks, 256>>>(d_pts, d, iters, 1); else k9l<3><<<blocks, 256>>>(d_pts, d, iters, 1);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best[v]) best[v] = ms;
    }
    const double adds = (double)blocks * 256 * iters;
    printf("{\"experiment\": \"carry passes of Q - X3 and 4p - Y1 folded into the fused double product (loose operands)\", \"waves_per_simd\": %d, \"carried_ms\": %.3f, \"loose_ms\": %.3f, "
           "\"carried_g_adds_per_s\": %.2f, \"loose_g_adds_per_s\": %.2f}\n", waves, best[0], best[1], adds / (best[0] * 1e-3) / 1e9, adds / (best[1] * 1e-3) / 1e9);
    (void)hipFree(d);
}
// the G2 loop body (g2_mmadd9: XYZZ over Fq2 on nine limbs), 2 waves per SIMD as k_msm_gather<G2Msm>
__global__ void __launch_bounds__(256, 2) k2(const uint32_t* pts, uint32_t* out, int iters) {
    g2_aff9 q[2];
    const uint32_t* src = pts + 40u * threadIdx.x;
    for (int j = 0; j < 2; j++) for (int k = 0; k < 9; k++) {
        q[j].x.c0.v[k] = src[j * 20 + k]; q[j].x.c1.v[k] = src[j * 20 + 10 + k]; q[j].y.c0.v[k] = src[40 * 7 + j * 20 + k]; q[j].y.c1.v[k] = src[40 * 7 + j * 20 + 10 + k];
    }
    g2_xyzz9 acc{q[0].x, q[0].y, q[1].x, q[1].y};
    for (int it = 0; it < iters; it++) acc = g2_mmadd9(acc, q[it & 1], (it >> 1) & 1);
    uint32_t s = 0; const fq9* c = reinterpret_cast<const fq9*>(&acc);
    for (int t = 0; t < 8; t++) for (int k = 0; k < 9; k++) s ^= c[t].v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static void run_g2(int waves, const uint32_t* d_pts) {
    const int blocks = 256 * waves, iters = 200;
    uint32_t* d; (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 4; r++) { (void)hipEventRecord(e0); k2<<<blocks, 256>>>(d_pts, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    const double adds = (double)blocks * 256 * iters;
    printf("{\"form\": \"G2 9x29\", \"operands\": \"per lane\", \"waves_per_simd\": %d, \"ms\": %.3f, \"g_adds_per_s\": %.2f, \"t_mad_per_s\": %.2f}\n", waves, best, adds / (best * 1e-3) / 1e9, adds * 4536 / (best * 1e-3) / 1e12);
    (void)hipFree(d);
}
template <int MAXW>
__global__ void __launch_bounds__(256, MAXW) k(const uint32_t* pts, uint32_t* out, int iters) {
    g1_aff q[2];
    for (int j = 0; j < 2; j++) for (int k = 0; k < 10; k++) { q[j].x.v[k] = pts[j * 20 + k]; q[j].y.v[k] = pts[j * 20 + 10 + k]; }
    g1_xyzz acc; acc.X = q[0].x; acc.Y = q[0].y; acc.ZZ = fq_one(); acc.ZZZ = fq_one();
    acc.X.v[0] ^= threadIdx.x & 1u;                       // not on the curve, which the formulas do not care about
    for (int it = 0; it < iters; it++) {
        g1_aff e = q[it & 1];
        e.y = fq_select((it >> 1) & 1, fq_sub_k4(fq_zero(), e.y), e.y);
        acc = g1_mmadd_lazy(acc, e);
    }
    uint32_t s = 0; for (int k = 0; k < 10; k++) s ^= acc.X.v[k] ^ acc.Y.v[k] ^ acc.ZZ.v[k] ^ acc.ZZZ.v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MAXW, bool NINE = false> void run(int waves, const uint32_t* d_pts, int per_lane = 0) {
    const int blocks = 256 * waves, iters = 400;
    uint32_t* d; (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 4; r++) { (void)hipEventRecord(e0); if (NINE) k9<MAXW><<<blocks, 256>>>(d_pts, d, iters, per_lane); else k<MAXW><<<blocks, 256>>>(d_pts, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    const double adds = (double)blocks * 256 * iters;
    printf("{\"form\": \"%s\", \"operands\": \"%s\", \"launch_bounds_waves\": %d, \"waves_per_simd\": %d, \"ms\": %.3f, \"g_adds_per_s\": %.2f, \"t_mad_per_s\": %.2f}\n", NINE ? "9x29" : "10x26", per_lane ? "per lane" : "same in every lane", MAXW, waves, best, adds / (best * 1e-3) / 1e9, adds * (NINE ? 1629 : 1810) / (best * 1e-3) / 1e12);
    (void)hipFree(d);
}
// `g1_add_rate sustain [seconds]`: the nine-limb loop at 3 waves per SIMD launched back to back for `seconds` (default 3): the rate a
// launch reaches after the chip has been under this load for a while, against the best-of-four of a cold 5 ms launch that the table above
// reports (a power-limited chip clocks down under sustained integer-multiply load: the ceiling a long-running prover sees is this one).
static int sustain(double seconds, const uint32_t* d_pts) {
    const int blocks = 256 * 3, iters = 400;
    uint32_t* d; (void)hipMalloc(&d, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const double adds = (double)blocks * 256 * iters;
    double elapsed = 0; int n = 0; float first = 0, last = 0, lo = 1e30f, hi = 0;
    while (elapsed < seconds * 1e3) {
        (void)hipEventRecord(e0); k9<3><<<blocks, 256>>>(d_pts, d, iters, 1); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (n == 0) first = ms;
        if (elapsed > seconds * 500.0) { if (ms < lo) lo = ms; if (ms > hi) hi = ms; }
        last = ms; elapsed += ms; n++;
    }
    printf("{\"mode\": \"sustained\", \"form\": \"9x29\", \"waves_per_simd\": 3, \"launches\": %d, \"seconds\": %.2f, \"first_ms\": %.3f, \"last_ms\": %.3f, \"second_half_min_ms\": %.3f, \"second_half_max_ms\": %.3f, "
           "\"g_adds_per_s_first\": %.2f, \"g_adds_per_s_last\": %.2f, \"g_adds_per_s_average\": %.2f}\n", n, elapsed / 1e3, first, last, lo, hi,
           adds / (first * 1e-3) / 1e9, adds / (last * 1e-3) / 1e9, adds * n / (elapsed * 1e-3) / 1e9);
    (void)hipFree(d);
    return 0;
}
int main(int argc, char** argv) {
    static uint32_t h[40 * 256]; uint64_t s = 0x9E3779B97F4A7C15ull;
    for (int k = 0; k < 40; k++) h[k] = 0x1234567u * (k + 1) & 0x3ffffffu;      // < 2^26: carried in either form
    for (int k = 40; k < 40 * 256; k++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[k] = (uint32_t)s & 0x3ffffffu; }
    uint32_t* d_pts; (void)hipMalloc(&d_pts, sizeof h); (void)hipMemcpy(d_pts, h, sizeof h, hipMemcpyHostToDevice);
    if (argc > 1 && argv[1][0] == 's') return sustain(argc > 2 ? atof(argv[2]) : 3.0, d_pts);
    if (argc > 1 && argv[1][0] == 'l') { run_loose(2, d_pts); run_loose(3, d_pts); return 0; }
    for (int w : {1, 2, 3}) run<3>(w, d_pts);            // 168-VGPR budget, as the MSM kernel
    for (int w : {2, 4}) run<4>(w, d_pts);               // 128-VGPR budget
    for (int w : {1, 2}) run<2>(w, d_pts);               // 256-VGPR budget
    for (int w : {1, 2, 3}) run<3, true>(w, d_pts);
    for (int w : {2, 3}) run<3, true>(w, d_pts, 1);      // operands that differ from lane to lane
    for (int w : {1, 2}) run_g2(w, d_pts);      // nine 29-bit limbs: 7 x 162 + 2 x 126 + 243 = 1 629 multiply-adds per addition
    return 0;
}
