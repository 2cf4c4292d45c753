// Groth16 verification through the Fq2 virtual machine (sixth translation unit of libzkp_hip; fq2vm.h, tools/gen_fq2vm.py).
// k_g16_public_inputs (16 lanes per envelope) accumulates the public inputs; k_g16_pairs_vm (lane = envelope) parses the points and lays the three (Q, P) pairs out as the
// machine's slot buffer; k_fq2vm runs a chain for 32 envelopes per workgroup on four cooperating waves -- chain A: the Miller loop of
// (B, A); chain B: the Miller loops of (gamma, -L) and (delta, -C) on the key's line table, one shared accumulator; the subgroup check of
// B; the final exponentiation -- and k_g16_vm_verdict reads the results.  A, B and the subgroup check run side by side.  Envelopes with a point
// at infinity in the proof (valid encodings that drop a pair from the product) are counted and left to the lane-per-chain kernels.
#include <hip/hip_runtime.h>
#include <vector>
#include "g16_rlc.h"
#include "fq2vm.h"
#include "g16_verify_launch.h"
using namespace zkp;

__global__ void __launch_bounds__(256) k_fq2vm(fq2vm::Launch L) {
    extern __shared__ uint32_t fq2vm_regs[];
    if (gridDim.x == 1) __builtin_amdgcn_s_setprio(3);          // a one-workgroup launch is somebody's critical path (the batch check's virtual envelope): issue before the co-resident waves
    fq2vm::run_device(L, fq2vm_regs);
}

__device__ inline void vm_put(uint32_t* io, uint32_t n, uint32_t i, uint32_t slot, const fq2& x) {
    for (uint32_t k = 0; k < 10; k++) { io[((size_t)slot * fq2vm::FQ2_W + k) * n + i] = x.c0.v[k]; io[((size_t)slot * fq2vm::FQ2_W + 10 + k) * n + i] = x.c1.v[k]; }
}
__device__ inline fq2 vm_get(const uint32_t* io, uint32_t n, uint32_t i, uint32_t slot) {
    fq2 x; for (uint32_t k = 0; k < 10; k++) { x.c0.v[k] = io[((size_t)slot * fq2vm::FQ2_W + k) * n + i]; x.c1.v[k] = io[((size_t)slot * fq2vm::FQ2_W + 10 + k) * n + i]; }
    return x;
}
// The public-input point L of every envelope, PI_LANES lanes per envelope (g16_verify.h: g16_public_input_lane), the lanes' partial points
// joined by a tree through LDS: 4 envelopes per 64-lane workgroup, so 1024 membership envelopes are 256 waves instead of 16 and an envelope's
// chain of additions is ~10 table steps + 4 tree levels instead of ~170.  Lbuf: [30 words][n] Jacobian (infinity for a malformed envelope,
// which k_g16_pairs_vm rejects from the same header check).
constexpr uint32_t PI_LANES = 16, PI_ENV = 64 / PI_LANES;
__global__ void __launch_bounds__(64) k_g16_public_inputs(int kind, const uint8_t* in, uint64_t stride, const uint32_t* len, uint32_t n, G16Vk vk, uint32_t* Lbuf) {
    __shared__ uint32_t sh[30 * 64];
    const uint32_t t = threadIdx.x, i = blockIdx.x * PI_ENV + t / PI_LANES, lane = t % PI_LANES;
    g1_jac acc = jac_infinity<fq>();
    if (i < n) {
        const uint32_t l = len[i] <= stride ? len[i] : 0u;
        G16Inputs h;
        if (g16_header(kind, vk, in + (uint64_t)i * stride, l, h)) acc = g16_public_input_lane(vk, h, lane, PI_LANES);
    }
    for (uint32_t s = 1; s < PI_LANES; s <<= 1) {
        for (uint32_t k = 0; k < 10; k++) { sh[k * 64 + t] = acc.X.v[k]; sh[(10 + k) * 64 + t] = acc.Y.v[k]; sh[(20 + k) * 64 + t] = acc.Z.v[k]; }
        __syncthreads();
        if ((lane & (2 * s - 1)) == 0) {
            g1_jac o;
            for (uint32_t k = 0; k < 10; k++) { o.X.v[k] = sh[k * 64 + t + s]; o.Y.v[k] = sh[(10 + k) * 64 + t + s]; o.Z.v[k] = sh[(20 + k) * 64 + t + s]; }
            acc = jac_add(acc, o);
        }
        __syncthreads();
    }
    if (lane == 0 && i < n) for (uint32_t k = 0; k < 10; k++) { Lbuf[(size_t)k * n + i] = acc.X.v[k]; Lbuf[(size_t)(10 + k) * n + i] = acc.Y.v[k]; Lbuf[(size_t)(20 + k) * n + i] = acc.Z.v[k]; }
}
// flags: 0 = a point fails to parse (verdict: reject), 1 = the generic case (all three pairs present, B finite), 2 = special
__global__ void __launch_bounds__(64) k_g16_pairs_vm(int kind, const uint8_t* in, uint64_t stride, const uint32_t* len, uint32_t n, G16Vk vk, const uint32_t* Lbuf, uint32_t* io, uint8_t* flags) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint32_t l = len[i] <= stride ? len[i] : 0u;
    const uint8_t* env = in + (uint64_t)i * stride;
    G16Pairs o;
    g1_jac L;                                             // the public-input point stays Jacobian: chain B takes it projectively (g16_vm_pair1)
    g1_jac Lin;
    for (uint32_t k = 0; k < 10; k++) { Lin.X.v[k] = Lbuf[(size_t)k * n + i]; Lin.Y.v[k] = Lbuf[(size_t)(10 + k) * n + i]; Lin.Z.v[k] = Lbuf[(size_t)(20 + k) * n + i]; }
    const bool v = kind == G16_EQUALITY ? g16_equality_pairs(vk, env, l, o, &L, &Lin) : g16_membership_pairs(vk, env, l, o, &L, &Lin);
    const bool generic = v && o.present == 15u;
    flags[i] = !v ? 0 : generic ? 1 : 2;
    for (uint32_t j = 0; j < 3; j++) {
        fq2 qx, qy, p;
        if (generic && j == 1) { g16_vm_pair1(L, p, qx); f_set_zero(qy); }          // chain B reads gamma's lines from the key's table: the Q.x slot carries Z^3
        else if (generic) { qx = o.Q[j].x; qy = o.Q[j].y; p = fq2{o.P[j].x, o.P[j].y}; }
        else { f_set_zero(qx); f_set_zero(qy); f_set_zero(p); }
        vm_put(io, n, i, fq2vm::PAIR_SLOTS * j + fq2vm::SLOT_QX, qx);
        vm_put(io, n, i, fq2vm::PAIR_SLOTS * j + fq2vm::SLOT_QY, qy);
        vm_put(io, n, i, fq2vm::PAIR_SLOTS * j + fq2vm::SLOT_P, p);
    }
}
__global__ void __launch_bounds__(64) k_g16_vm_verdict(uint32_t n, const uint32_t* io, const uint8_t* flags, uint8_t* ok, uint32_t* special) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    if (flags[i] != 1) { ok[i] = 0; if (flags[i] == 2) atomicAdd(special, 1u); return; }
    // B in the subgroup: [6x^2] B is finite and equals psi(B) (tools/gen_fq2vm.py S_LAST: Z, and the differences of the two coordinates)
    bool good = !f_is_zero(vm_get(io, n, i, fq2vm::SLOT_SZ)) && f_is_zero(vm_get(io, n, i, fq2vm::SLOT_SH)) && f_is_zero(vm_get(io, n, i, fq2vm::SLOT_SR));
    good = good && fq2_eq(vm_get(io, n, i, fq2vm::SLOT_RES), fq2_one());
    for (uint32_t k = 1; k < 6; k++) good = good && f_is_zero(vm_get(io, n, i, fq2vm::SLOT_RES + k));
    ok[i] = good ? 1 : 0;
}

size_t g16_vm_scratch_bytes(uint32_t n) { return (size_t)n * ((size_t)fq2vm::N_SLOTS * fq2vm::FQ2_W * 4 + 30 * 4 + 1) + 512; }          // the slot buffer, the public-input points, then one flag byte per envelope

int g16_vm_upload(G16VmTables& T) {          // (g16_vm_free is declared in g16_verify_launch.h)
    auto up = [](const void* src, size_t bytes, const void** dst) -> int {
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) return -1;
        if (hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p); return -1; }
        *dst = p; return 0;
    };
    int rc = 0;
    rc |= up(fq2vm::CODE_K4, sizeof(fq2vm::CODE_K4), (const void**)&T.code); rc |= up(fq2vm::OFF_K4, sizeof(fq2vm::OFF_K4), (const void**)&T.off);
    rc |= up(fq2vm::CONSTS, sizeof(fq2vm::CONSTS), (const void**)&T.consts);
    rc |= up(fq2vm::SCRIPT_MILLER, sizeof(fq2vm::SCRIPT_MILLER), (const void**)&T.script[0]);
    rc |= up(fq2vm::SCRIPT_SUBGROUP, sizeof(fq2vm::SCRIPT_SUBGROUP), (const void**)&T.script[1]);
    rc |= up(fq2vm::SCRIPT_FINISH, sizeof(fq2vm::SCRIPT_FINISH), (const void**)&T.script[2]);
    rc |= up(fq2vm::SCRIPT_MILLER_B, sizeof(fq2vm::SCRIPT_MILLER_B), (const void**)&T.script[3]);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_fq2vm), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
        (void)hipGetLastError(); g16_vm_free(T); return -2;          // the 160 KB LDS opt-in of the Fq2 machine was refused: say so here, not as an opaque launch error later
    }
    for (auto& q : T.side) if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess) rc = -1;
    {
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); greatest = 0; }
        for (auto& q : T.vq) if (hipStreamCreateWithPriority(&q, hipStreamNonBlocking, greatest) != hipSuccess) rc = -1;
    }
    for (auto& e : T.ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc = -1;
    {
        // mask bit b = CU b / 8 of XCD b mod 8 (tools/cumask_probe.hip): bits 0..7 are the first CU of every XCD
        uint32_t mask[8] = {0xffffff00u, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        if (hipExtStreamCreateWithCUMask(&T.sub, 8, mask) != hipSuccess) { (void)hipGetLastError(); T.sub = nullptr; if (hipStreamCreateWithFlags(&T.sub, hipStreamNonBlocking) != hipSuccess) rc = -1; }
        for (auto& m : mask) m = ~m;
        if (hipExtStreamCreateWithCUMask(&T.tail, 8, mask) != hipSuccess) { (void)hipGetLastError(); T.tail = nullptr; if (hipStreamCreateWithFlags(&T.tail, hipStreamNonBlocking) != hipSuccess) rc = -1; }
    }
    if (rc) { g16_vm_free(T); return -1; }          // a partial upload leaves nothing behind
    T.ready = true;
    return 0;
}
void g16_vm_free(G16VmTables& T) {
    if (T.code) (void)hipFree((void*)T.code);
    if (T.off) (void)hipFree((void*)T.off);
    for (auto q : T.side) if (q) (void)hipStreamDestroy(q);
    for (auto q : T.vq) if (q) (void)hipStreamDestroy(q);
    if (T.sub) (void)hipStreamDestroy(T.sub);
    if (T.tail) (void)hipStreamDestroy(T.tail);
    for (auto e : T.ev) if (e) (void)hipEventDestroy(e);
    if (T.consts) (void)hipFree((void*)T.consts);
    for (int k = 0; k < 4; k++) if (T.script[k]) (void)hipFree((void*)T.script[k]);
    T = G16VmTables();
}
// What the machine needs from a key, computed on the host through the same tables: its Miller value of (beta, -alpha) (chain A's script
// on one envelope) and the line table of gamma and delta that chain B reads ([step][gamma, delta][3 coefficients], chain L twice).
void g16_vm_key_constants(const g2_aff& beta, const g1_aff& neg_alpha, const g2_aff& gamma, const g2_aff& delta, uint32_t ml[6 * 20], std::vector<uint32_t>& lines) {
    namespace vm = fq2vm;
    const vm::Tables T{vm::CODE_K4, vm::OFF_K4, 4, &vm::CONSTS[0][0]};
    auto put = [&](std::vector<uint32_t>& io, uint32_t slot, const fq2& x) { vm::fq2_to_words(&io[(size_t)slot * vm::FQ2_W], x); };
    {
        std::vector<uint32_t> io((size_t)vm::N_SLOTS * vm::FQ2_W, 0u);
        put(io, vm::SLOT_QX, beta.x); put(io, vm::SLOT_QY, beta.y); put(io, vm::SLOT_P, fq2{neg_alpha.x, neg_alpha.y});
        vm::Launch L{T, vm::SCRIPT_MILLER, (uint32_t)(sizeof(vm::SCRIPT_MILLER) / 2), 1, io.data(), 0, nullptr};
        vm::run_host(L, 0, vm::REGS_K4[0]);
        for (uint32_t k = 0; k < 6 * vm::FQ2_W; k++) ml[k] = io[(size_t)vm::SLOT_F0 * vm::FQ2_W + k];
    }
    lines.assign((size_t)vm::N_LINE_SLOTS * vm::FQ2_W, 0u);
    for (int j = 0; j < 2; j++) {
        const g2_aff& q = j ? delta : gamma;
        std::vector<uint32_t> io((size_t)(vm::LINE_SLOT0 + vm::N_LINE_SLOTS) * vm::FQ2_W, 0u);
        put(io, vm::SLOT_QX, q.x); put(io, vm::SLOT_QY, q.y);
        vm::Launch L{T, vm::SCRIPT_LINES, (uint32_t)(sizeof(vm::SCRIPT_LINES) / 2), 1, io.data(), 0, nullptr};
        vm::run_host(L, 0, vm::REGS_K4[4]);
        for (uint32_t st = 0; st < vm::N_LINE_SLOTS / 6; st++)
            for (uint32_t k = 0; k < 3 * vm::FQ2_W; k++) lines[((size_t)6 * st + 3 * j) * vm::FQ2_W + k] = io[((size_t)vm::LINE_SLOT0 + 6 * st) * vm::FQ2_W + k];
    }
}

void g16_launch_verify_vm(int kind, const uint8_t* d_in, uint64_t stride, const uint32_t* d_len, uint32_t n, const G16Vk& vk, const G16VmTables& T, const uint32_t* d_kconst,
                          const uint32_t* d_lines, void* d_scratch, uint8_t* d_ok, uint32_t* d_special, hipStream_t st) {
    if (!n) return;
    uint32_t* io = reinterpret_cast<uint32_t*>(d_scratch);
    uint32_t* Lbuf = io + (size_t)fq2vm::N_SLOTS * fq2vm::FQ2_W * n;
    uint8_t* flags = reinterpret_cast<uint8_t*>(Lbuf + (size_t)30 * n);
    const uint32_t nb = (n + 63) / 64;
    k_g16_public_inputs<<<(n + PI_ENV - 1) / PI_ENV, 64, 0, st>>>(kind, d_in, stride, d_len, n, vk, Lbuf);
    k_g16_pairs_vm<<<nb, 64, 0, st>>>(kind, d_in, stride, d_len, n, vk, Lbuf, io, flags);
    const uint32_t K = 4, ng = (n + fq2vm::G - 1) / fq2vm::G;
    const size_t pair_words = (size_t)fq2vm::PAIR_SLOTS * fq2vm::FQ2_W * n;
    // chain: 0 = A, 1 = subgroup, 2 = finish, 3 = B (the index of its script in T.script and of its register count in REGS_K4)
    auto launch = [&](int chain, uint32_t* base, const uint32_t* kconst, uint32_t script_len, hipStream_t s) {
        fq2vm::Launch L{{T.code, T.off, K, T.consts}, T.script[chain], script_len, n, base, 0, kconst};
        k_fq2vm<<<ng, K * 64, (size_t)fq2vm::REGS_K4[chain] * fq2vm::FQ2_W * fq2vm::G * 4, s>>>(L);
    };
    // Chain B runs beside chain A on a stream of its own (a workgroup's LDS: A 118 KB, B 115 KB: one per CU, 128 + 128 workgroups per 4096
    // envelopes).  The subgroup check of B (38 KB) only feeds the verdict and starts when chain A is done, beside the final exponentiation
    // (159 KB per workgroup: it cannot share a CU with anything, the subgroup workgroups take the other CUs): launched together with A and
    // B its waves shared the SIMDs of chain B's CUs and stretched B, which the final exponentiation waits for, from 3.4 to 5.2 ms.
    (void)hipEventRecord(T.ev[0], st);
    (void)hipStreamWaitEvent(T.side[0], T.ev[0], 0);
    launch(3, io + pair_words, d_lines, (uint32_t)(sizeof(fq2vm::SCRIPT_MILLER_B) / 2), T.side[0]);
    (void)hipEventRecord(T.ev[1], T.side[0]);
    launch(0, io, nullptr, (uint32_t)(sizeof(fq2vm::SCRIPT_MILLER) / 2), st);
    (void)hipEventRecord(T.ev[0], st);
    (void)hipStreamWaitEvent(T.side[1], T.ev[0], 0);
    launch(1, io, nullptr, (uint32_t)(sizeof(fq2vm::SCRIPT_SUBGROUP) / 2), T.side[1]);
    (void)hipEventRecord(T.ev[2], T.side[1]);
    (void)hipStreamWaitEvent(st, T.ev[1], 0);
    launch(2, io, d_kconst, (uint32_t)(sizeof(fq2vm::SCRIPT_FINISH) / 2), st);
    (void)hipStreamWaitEvent(st, T.ev[2], 0);
    k_g16_vm_verdict<<<nb, 64, 0, st>>>(n, io, flags, d_ok, d_special);
}

// ================================================================================================ one pairing check per batch (g16_rlc.h)
namespace {
constexpr uint32_t RLC_TERM_BLOCKS = 32, RLC_SUM_BLOCKS = 64, RLC_PROD_BLOCKS = 256;
enum { RLC_SPECIAL = 0, RLC_BAD_SUBGROUP = 1, RLC_ANOMALY = 2, RLC_BATCH_OK = 3, RLC_COUNTERS = 4 };
__device__ inline void st_jac(uint32_t* p, size_t n, size_t i, const g1_jac& a) { for (uint32_t k = 0; k < 10; k++) { p[k * n + i] = a.X.v[k]; p[(10 + k) * n + i] = a.Y.v[k]; p[(20 + k) * n + i] = a.Z.v[k]; } }
__device__ inline g1_jac ld_jac(const uint32_t* p, size_t n, size_t i) { g1_jac a; for (uint32_t k = 0; k < 10; k++) { a.X.v[k] = p[k * n + i]; a.Y.v[k] = p[(10 + k) * n + i]; a.Z.v[k] = p[(20 + k) * n + i]; } return a; }
// tree sum of one Jacobian point per lane of a 256-lane block through LDS (sh: 30 * 256 words); the total ends in lane 0
__device__ inline g1_jac block_sum_jac(g1_jac acc, uint32_t* sh) {
    const uint32_t t = threadIdx.x;
    for (uint32_t s = 1; s < 256; s <<= 1) {
        st_jac(sh, 256, t, acc);
        __syncthreads();
        if ((t & (2 * s - 1)) == 0) acc = jac_add(acc, ld_jac(sh, 256, t + s));
        __syncthreads();
    }
    return acc;
}
}  // namespace

// lane = envelope: parse; B_j into chain A's / the subgroup chain's slots, A_j and C_j (affine) into ACbuf ([2][20 words][n]).
// flags: 0 refused, 1 live, 2 a point at infinity
__global__ void __launch_bounds__(64) k_g16_rlc_parse(int kind, const uint8_t* in, uint64_t stride, const uint32_t* len, uint32_t n, G16Vk vk, uint32_t* io, uint32_t* ACbuf, uint8_t* flags) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint32_t l = len[i] <= stride ? len[i] : 0u;
    G16Inputs h;
    g1_aff A, C; g2_aff B;
    int ra = 0, rb = 0, rc = 0;
    const bool head = g16_header(kind, vk, in + (uint64_t)i * stride, l, h);
    if (head) { ra = g1_from_ark(A, h.proof); rb = g2_from_ark(B, h.proof + 64); rc = g1_from_ark(C, h.proof + 192); }
    const bool valid = head && ra && rb && rc, live = valid && ra == 1 && rb == 1 && rc == 1;
    flags[i] = !valid ? 0 : live ? 1 : 2;
    fq2 qx, qy; f_set_zero(qx); f_set_zero(qy);
    if (live) { qx = B.x; qy = B.y; }
    vm_put(io, n, i, fq2vm::SLOT_QX, qx); vm_put(io, n, i, fq2vm::SLOT_QY, qy);
    if (live) for (uint32_t k = 0; k < 10; k++) {
        ACbuf[(size_t)k * n + i] = A.x.v[k]; ACbuf[(size_t)(10 + k) * n + i] = A.y.v[k];
        ACbuf[(size_t)(20 + k) * n + i] = C.x.v[k]; ACbuf[(size_t)(30 + k) * n + i] = C.y.v[k];
    }
}
// lane = (envelope, blockIdx.y: 0 = A, 1 = C): the weighted point (g1_mul_weight).  w A goes to chain A's P slot as an affine point, w C to
// Cbuf as it is (only their sum is needed)
__global__ void __launch_bounds__(64) k_g16_rlc_mul(uint32_t n, const uint32_t* rho, const uint32_t* ACbuf, const uint8_t* flags, uint32_t* io, uint32_t* Cbuf) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x, which = blockIdx.y;
    if (i >= n) return;
    g1_jac r = jac_infinity<fq>();
    if (flags[i] == 1) {
        g1_aff P; for (uint32_t k = 0; k < 10; k++) { P.x.v[k] = ACbuf[(size_t)(20 * which + k) * n + i]; P.y.v[k] = ACbuf[(size_t)(20 * which + 10 + k) * n + i]; }
        const uint32_t w[4] = {rho[4 * i], rho[4 * i + 1], rho[4 * i + 2], rho[4 * i + 3]};
        r = g1_mul_weight(P, w);
    }
    if (which == 1) { st_jac(Cbuf, n, i, r); return; }
    fq2 p; f_set_zero(p);
    g1_aff a;
    if (jac_to_aff(a, r)) p = fq2{a.x, a.y};                         // (finite for a live envelope: A is, the weight is not 0 mod r, the group has prime order)
    vm_put(io, n, i, fq2vm::SLOT_P, p);
}
// S_t = sum over live envelopes of rho_j x_{j,t}: blockIdx.y = t, the blocks of a row stride over the envelopes; part[t][block] (Montgomery form)
__global__ void __launch_bounds__(256) k_g16_rlc_terms(int kind, const uint8_t* in, uint64_t stride, const uint32_t* len, uint32_t n, G16Vk vk, const uint32_t* rho, const uint8_t* flags, uint32_t* part) {
    __shared__ uint32_t sh[8 * 256];
    const uint32_t t = threadIdx.x, term = blockIdx.y;
    fr acc = fp_zero<FrParams>();
    for (uint32_t j = blockIdx.x * 256 + t; j < n; j += gridDim.x * 256) {
        if (flags[j] != 1) continue;
        G16Inputs h;
        if (!g16_header(kind, vk, in + (uint64_t)j * stride, len[j], h)) continue;          // (live envelopes passed it already)
        fr x;
        if (g16_rlc_term(h, term, g16_rlc_weight(rho + 4 * j), x)) acc = fp_add(acc, x);
    }
    for (uint32_t s = 1; s < 256; s <<= 1) {
        for (uint32_t k = 0; k < 8; k++) sh[k * 256 + t] = acc.v[k];
        __syncthreads();
        if ((t & (2 * s - 1)) == 0) { fr o; for (uint32_t k = 0; k < 8; k++) o.v[k] = sh[k * 256 + t + s]; acc = fp_add(acc, o); }
        __syncthreads();
    }
    if (t == 0) for (uint32_t k = 0; k < 8; k++) part[((size_t)term * gridDim.x + blockIdx.x) * 8 + k] = acc.v[k];
}
// lane t <= n_ic: canonical words of S_t (t < n_ic) or of 1 - S_0 (t = n_ic: the scalar of alpha)
__global__ void __launch_bounds__(256) k_g16_rlc_scalars(uint32_t n_ic, uint32_t nblk, const uint32_t* part, uint32_t* scal) {
    const uint32_t t = threadIdx.x;
    if (t > n_ic) return;
    const uint32_t src = t < n_ic ? t : 0u;
    fr acc = fp_zero<FrParams>();
    for (uint32_t b = 0; b < nblk; b++) { fr o; for (uint32_t k = 0; k < 8; k++) o.v[k] = part[((size_t)src * nblk + b) * 8 + k]; acc = fp_add(acc, o); }
    if (t == n_ic) acc = g16_rlc_one_minus(acc);
    uint32_t w[8]; fp_to_raw(w, acc);
    for (uint32_t k = 0; k < 8; k++) scal[(size_t)t * 8 + k] = w[k];
}
// block 0: L_V = sum_t S_t IC_t; block 1: A_V = (1 - S_0) alpha (point n_ic of the window tables).  256 lanes share the (point, window) steps.
__global__ void __launch_bounds__(256) k_g16_rlc_fixed(G16Vk vk, const uint32_t* scal, uint32_t* pts) {
    __shared__ uint32_t sh[30 * 256];
    const uint32_t first = blockIdx.x == 0 ? 0u : vk.n_ic, count = blockIdx.x == 0 ? vk.n_ic : 1u;
    g1_jac acc = jac_infinity<fq>();
    for (uint32_t s = threadIdx.x; s < count * G16V_NWIN; s += 256) {
        const uint32_t ic = first + s / G16V_NWIN, w = s % G16V_NWIN;
        uint32_t k[8]; for (uint32_t j = 0; j < 8; j++) k[j] = scal[(size_t)ic * 8 + j];
        const int32_t d = g16_ic_digit(k, w);
        if (d == 0) continue;
        const uint32_t* e = vk.ic_table + (((size_t)ic * G16V_NWIN + w) * G16V_NENT + (uint32_t)((d < 0 ? -d : d) - 1)) * 20;
        g1_aff q; for (int j = 0; j < 10; j++) { q.x.v[j] = e[j]; q.y.v[j] = e[10 + j]; }
        if (d < 0) q.y = fq_neg(q.y);
        acc = jac_madd(acc, q);
    }
    acc = block_sum_jac(acc, sh);
    if (threadIdx.x == 0) st_jac(pts + 30 * blockIdx.x, 1, 0, acc);
}
// sums of Jacobian points: block b adds in[b * 256 + t + k * 256 * gridDim.x] for all k, then the tree; out[b] (out_n = row length of `out`)
__global__ void __launch_bounds__(256) k_g1_sum(const uint32_t* in, uint32_t n_in, uint32_t* out, uint32_t out_n) {
    __shared__ uint32_t sh[30 * 256];
    g1_jac acc = jac_infinity<fq>();
    for (uint32_t j = blockIdx.x * 256 + threadIdx.x; j < n_in; j += gridDim.x * 256) acc = jac_add(acc, ld_jac(in, n_in, j));
    acc = block_sum_jac(acc, sh);
    if (threadIdx.x == 0) st_jac(out, out_n, blockIdx.x, acc);
}
// the virtual envelope's slots (one lane).  part 0: pair 0 = (beta, A_V); part 1: pair 1 = L_V in chain B's projective form, pair 2 = -C_V
__global__ void __launch_bounds__(64) k_g16_rlc_virtual(G16Vk vk, const uint32_t* pts, uint32_t* iov, uint32_t* counters, uint32_t part) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const uint32_t P = fq2vm::PAIR_SLOTS;
    if (part == 0) {
        g1_aff A;
        if (!jac_to_aff(A, ld_jac(pts + 30, 1, 0))) { atomicAdd(counters + RLC_ANOMALY, 1u); return; }          // (probability ~2^-128; the caller then verifies envelope by envelope)
        vm_put(iov, 1, 0, fq2vm::SLOT_QX, vk.beta.x); vm_put(iov, 1, 0, fq2vm::SLOT_QY, vk.beta.y); vm_put(iov, 1, 0, fq2vm::SLOT_P, fq2{A.x, A.y});
        return;
    }
    const g1_jac Lv = ld_jac(pts, 1, 0);
    g1_aff C;
    if (!jac_to_aff(C, ld_jac(pts + 60, 1, 0)) || jac_is_inf(Lv)) { atomicAdd(counters + RLC_ANOMALY, 1u); return; }
    fq2 p1, p1z, zero; f_set_zero(zero);
    g16_vm_pair1(Lv, p1, p1z);
    vm_put(iov, 1, 0, P + fq2vm::SLOT_QX, p1z); vm_put(iov, 1, 0, P + fq2vm::SLOT_QY, zero); vm_put(iov, 1, 0, P + fq2vm::SLOT_P, p1);
    const g1_aff nC = aff_neg(C);
    vm_put(iov, 1, 0, 2 * P + fq2vm::SLOT_QX, vk.delta.x); vm_put(iov, 1, 0, 2 * P + fq2vm::SLOT_QY, vk.delta.y); vm_put(iov, 1, 0, 2 * P + fq2vm::SLOT_P, fq2{nC.x, nC.y});
}
// products of chain-A values: `f` = the six value slots of a slot buffer of n_in envelopes ([6 * 20 words][n_in]); block b multiplies the
// values of envelopes b * 64 + t + k * 64 * gridDim.x whose flag is 1 (flags == nullptr: all), tree through LDS, out[b] in the same layout with
// row length out_n.  extra != nullptr (one-envelope buffer): block 0 multiplies its value in as well.
__global__ void __launch_bounds__(64) k_fq12_prod(const uint32_t* f, uint32_t n_in, const uint8_t* flags, uint32_t* out, uint32_t out_n, const uint32_t* extra) {
    __shared__ uint32_t sh[120 * 64];
    const uint32_t t = threadIdx.x;
    auto ld = [](const uint32_t* p, size_t n, size_t i) { fq2 c[6]; for (uint32_t s = 0; s < 6; s++) for (uint32_t k = 0; k < 10; k++) { c[s].c0.v[k] = p[(s * 20 + k) * n + i]; c[s].c1.v[k] = p[(s * 20 + 10 + k) * n + i]; } return fq12_from_coeffs(c); };
    auto st = [](uint32_t* p, size_t n, size_t i, const fq12& x) { fq2 c[6]; fq12_to_coeffs(c, x); for (uint32_t s = 0; s < 6; s++) for (uint32_t k = 0; k < 10; k++) { p[(s * 20 + k) * n + i] = c[s].c0.v[k]; p[(s * 20 + 10 + k) * n + i] = c[s].c1.v[k]; } };
    fq12 acc = fq12_one();
    bool any = false;
    for (uint32_t j = blockIdx.x * 64 + t; j < n_in; j += gridDim.x * 64) {
        if (flags != nullptr && flags[j] != 1) continue;
        const fq12 x = ld(f, n_in, j);
        acc = any ? fq12_mul(acc, x) : x; any = true;
    }
    if (extra != nullptr && blockIdx.x == 0 && t == 0) { const fq12 x = ld(extra, 1, 0); acc = any ? fq12_mul(acc, x) : x; any = true; }
    for (uint32_t s = 1; s < 64; s <<= 1) {
        st(sh, 64, t, acc);
        __syncthreads();
        if ((t & (2 * s - 1)) == 0) acc = fq12_mul(acc, ld(sh, 64, t + s));
        __syncthreads();
    }
    if (t == 0) st(out, out_n, blockIdx.x, acc);
}
// verdicts of the batch: ok[i] = live and B_i in the subgroup; counters say whether they stand (no special envelope, no live envelope outside the
// subgroup, no anomaly, and the virtual envelope's check -- the whole product -- equal to one)
__global__ void __launch_bounds__(64) k_g16_rlc_verdict(uint32_t n, const uint32_t* io, const uint8_t* flags, const uint32_t* iov, uint8_t* ok, uint32_t* counters) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i == 0) {
        bool good = fq2_eq(vm_get(iov, 1, 0, fq2vm::SLOT_RES), fq2_one());
        for (uint32_t k = 1; k < 6; k++) good = good && f_is_zero(vm_get(iov, 1, 0, fq2vm::SLOT_RES + k));
        counters[RLC_BATCH_OK] = good ? 1u : 0u;
    }
    if (i >= n) return;
    if (flags[i] != 1) { ok[i] = 0; if (flags[i] == 2) atomicAdd(counters + RLC_SPECIAL, 1u); return; }
    const bool sub = !f_is_zero(vm_get(io, n, i, fq2vm::SLOT_SZ)) && f_is_zero(vm_get(io, n, i, fq2vm::SLOT_SH)) && f_is_zero(vm_get(io, n, i, fq2vm::SLOT_SR));
    if (!sub) atomicAdd(counters + RLC_BAD_SUBGROUP, 1u);
    ok[i] = sub ? 1 : 0;
}

namespace {
struct RlcLayout { size_t io, cbuf, acbuf, rho, part, scal, pts, iov, sums, prods, counters, flags, total; };
RlcLayout rlc_layout(uint32_t n, uint32_t n_ic) {
    RlcLayout L; size_t off = 0;
    auto words = [&](size_t w) { const size_t o = off; off += (w * 4 + 255) & ~(size_t)255; return o; };
    L.io = words((size_t)fq2vm::N_SLOTS * fq2vm::FQ2_W * n); L.cbuf = words((size_t)30 * n); L.acbuf = words((size_t)40 * n); L.rho = words((size_t)4 * n);
    L.part = words((size_t)n_ic * RLC_TERM_BLOCKS * 8); L.scal = words((size_t)(n_ic + 1) * 8); L.pts = words(90);
    L.iov = words((size_t)fq2vm::N_SLOTS * fq2vm::FQ2_W); L.sums = words((size_t)30 * RLC_SUM_BLOCKS); L.prods = words((size_t)120 * RLC_PROD_BLOCKS);
    L.counters = words(RLC_COUNTERS); L.flags = words((n + 3) / 4);
    L.total = off;
    return L;
}
}  // namespace
size_t g16_rlc_scratch_bytes(uint32_t n, uint32_t n_ic) { return rlc_layout(n, n_ic).total; }
size_t g16_rlc_rho_offset(uint32_t n, uint32_t n_ic) { return rlc_layout(n, n_ic).rho; }
size_t g16_rlc_counters_offset(uint32_t n, uint32_t n_ic) { return rlc_layout(n, n_ic).counters; }

// d_scratch: g16_rlc_scratch_bytes(n, vk.n_ic), with the weights (4 words per envelope, non-zero) already at g16_rlc_rho_offset and the
// counters zeroed.  Leaves ok[] and the four counters (g16_rlc_counters_offset: special, live-outside-subgroup, anomaly, batch-ok).
void g16_launch_verify_rlc(int kind, const uint8_t* d_in, uint64_t stride, const uint32_t* d_len, uint32_t n, const G16Vk& vk, const G16VmTables& T, const uint32_t* d_kconst,
                           const uint32_t* d_lines, void* d_scratch, uint8_t* d_ok, hipStream_t st) {
    if (!n) return;
    const RlcLayout Y = rlc_layout(n, vk.n_ic);
    uint8_t* base = reinterpret_cast<uint8_t*>(d_scratch);
    auto W = [&](size_t o) { return reinterpret_cast<uint32_t*>(base + o); };
    uint32_t *io = W(Y.io), *cbuf = W(Y.cbuf), *acbuf = W(Y.acbuf), *rho = W(Y.rho), *part = W(Y.part), *scal = W(Y.scal), *pts = W(Y.pts), *iov = W(Y.iov), *sums = W(Y.sums), *prods = W(Y.prods), *counters = W(Y.counters);
    uint8_t* flags = base + Y.flags;
    const uint32_t nb = (n + 63) / 64, K = 4;
    auto launch = [&](int chain, uint32_t count, uint32_t* buf, const uint32_t* kconst, uint32_t script_len, hipStream_t s) {
        fq2vm::Launch L{{T.code, T.off, K, T.consts}, T.script[chain], script_len, count, buf, 0, kconst};
        k_fq2vm<<<(count + fq2vm::G - 1) / fq2vm::G, K * 64, (size_t)fq2vm::REGS_K4[chain] * fq2vm::FQ2_W * fq2vm::G * 4, s>>>(L);
    };
    // Streams.  st: parse | weighted points | chain A of the batch | product of its values | x V's value | V's finishing chain | verdict.
    // side[1] / sub: the subgroup chain of every B_j (a single round of chain A: from the parse on, beside the weighted points, which use no
    // LDS; several rounds: after chain A, below).  vq[0] (greatest priority): the scalar sums,
    // V's fixed-base points, V's chain A.  vq[1] (greatest priority): the sum of the w C, V's chain B.  The chains' workgroups hold 118 / 41 KB of LDS, and
    // whatever is launched behind a full grid of them waits for a CU to drain: when the batch's chain A is a single round of workgroups
    // (n <= 32 per CU) it is launched AFTER V's short kernels and one-workgroup chains have been placed -- V is the critical path there;
    // a longer chain A starts as soon as the weighted points exist and V's work (~8 ms) hides beneath it.
    const bool a_first = (n + fq2vm::G - 1) / fq2vm::G > 256u;
    (void)hipMemsetAsync(iov, 0, (size_t)fq2vm::N_SLOTS * fq2vm::FQ2_W * 4, st);
    k_g16_rlc_parse<<<nb, 64, 0, st>>>(kind, d_in, stride, d_len, n, vk, io, acbuf, flags);
    (void)hipEventRecord(T.ev[0], st);
    auto subgroup_chain = [&](hipStream_t q) {
        (void)hipStreamWaitEvent(q, T.ev[0], 0);
        launch(1, n, io, nullptr, (uint32_t)(sizeof(fq2vm::SCRIPT_SUBGROUP) / 2), q);
        (void)hipEventRecord(T.ev[2], q);
    };
    if (!a_first) subgroup_chain(T.side[1]);
    (void)hipStreamWaitEvent(T.vq[0], T.ev[0], 0);
    k_g16_rlc_terms<<<dim3(RLC_TERM_BLOCKS, vk.n_ic), 256, 0, T.vq[0]>>>(kind, d_in, stride, d_len, n, vk, rho, flags, part);
    k_g16_rlc_scalars<<<1, 256, 0, T.vq[0]>>>(vk.n_ic, RLC_TERM_BLOCKS, part, scal);
    k_g16_rlc_fixed<<<2, 256, 0, T.vq[0]>>>(vk, scal, pts);
    (void)hipEventRecord(T.ev[4], T.vq[0]);
    k_g16_rlc_virtual<<<1, 64, 0, T.vq[0]>>>(vk, pts, iov, counters, 0u);
    launch(0, 1, iov, nullptr, (uint32_t)(sizeof(fq2vm::SCRIPT_MILLER) / 2), T.vq[0]);
    (void)hipEventRecord(T.ev[1], T.vq[0]);
    k_g16_rlc_mul<<<dim3(nb, 2), 64, 0, st>>>(n, rho, acbuf, flags, io, cbuf);
    (void)hipEventRecord(T.ev[0], st);
    if (a_first) launch(0, n, io, nullptr, (uint32_t)(sizeof(fq2vm::SCRIPT_MILLER) / 2), st);
    (void)hipStreamWaitEvent(T.vq[1], T.ev[0], 0);
    k_g1_sum<<<RLC_SUM_BLOCKS, 256, 0, T.vq[1]>>>(cbuf, n, sums, RLC_SUM_BLOCKS);
    k_g1_sum<<<1, 256, 0, T.vq[1]>>>(sums, RLC_SUM_BLOCKS, pts + 60, 1);
    (void)hipStreamWaitEvent(T.vq[1], T.ev[4], 0);                 // (pts[0] = L_V comes from vq[0])
    k_g16_rlc_virtual<<<1, 64, 0, T.vq[1]>>>(vk, pts, iov, counters, 1u);
    launch(3, 1, iov + (size_t)fq2vm::PAIR_SLOTS * fq2vm::FQ2_W, d_lines, (uint32_t)(sizeof(fq2vm::SCRIPT_MILLER_B) / 2), T.vq[1]);
    (void)hipEventRecord(T.ev[3], T.vq[1]);
    if (!a_first) launch(0, n, io, nullptr, (uint32_t)(sizeof(fq2vm::SCRIPT_MILLER) / 2), st);
    const uint32_t* fin = io + (size_t)fq2vm::SLOT_F0 * fq2vm::FQ2_W * n;
    uint32_t* fv = iov + (size_t)fq2vm::SLOT_F0 * fq2vm::FQ2_W;
    // Several rounds of chain A: the subgroup chain runs AFTER it and the first product pass, beside the rest of the tail (V's finishing chain: ~4 ms on one CU; the subgroup chain's stream
    // leaves the first CU of every XCD free and the tail's stream is confined to those).  Beside chain A it only stretched it: both are bound by the SIMDs, not by LDS space -- 28.8 ms
    // together against 21.4 + 5.6 ms one after the other at 65 536 envelopes.  (Also measured: the subgroup chain of the first 17 920 envelopes from the
    // parse on, beside the weighted points, the rest after chain A -- the early part stretched the weighted points from 2.0 to 4.2 ms and the call
    // took 35.75 against 35.9 ms: the same work on the same SIMDs in another order.)
    k_fq12_prod<<<RLC_PROD_BLOCKS, 64, 0, st>>>(fin, n, flags, prods, RLC_PROD_BLOCKS, nullptr);
    hipStream_t tq = st;
    if (a_first) { (void)hipEventRecord(T.ev[0], st); subgroup_chain(T.sub); tq = T.tail; (void)hipStreamWaitEvent(tq, T.ev[0], 0); }
    (void)hipStreamWaitEvent(tq, T.ev[1], 0);
    (void)hipStreamWaitEvent(tq, T.ev[3], 0);
    k_fq12_prod<<<1, 64, 0, tq>>>(prods, RLC_PROD_BLOCKS, nullptr, fv, 1, fv);
    launch(2, 1, iov, d_kconst, (uint32_t)(sizeof(fq2vm::SCRIPT_FINISH) / 2), tq);
    if (a_first) { (void)hipEventRecord(T.ev[5], tq); (void)hipStreamWaitEvent(st, T.ev[5], 0); }
    (void)hipStreamWaitEvent(st, T.ev[2], 0);
    k_g16_rlc_verdict<<<nb, 64, 0, st>>>(n, io, flags, iov, d_ok, counters);
}
