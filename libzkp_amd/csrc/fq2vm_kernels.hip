// Groth16 verification through the Fq2 virtual machine (sixth translation unit of libzkp_hip; fq2vm.h, tools/gen_fq2vm.py).
// k_g16_public_inputs (16 lanes per envelope) accumulates the public inputs; k_g16_pairs_vm (lane = envelope) parses the points and lays the three (Q, P) pairs out as the
// machine's slot buffer; k_fq2vm runs a chain for 32 envelopes per workgroup on four cooperating waves -- chain A: the Miller loop of
// (B, A); chain B: the Miller loops of (gamma, -L) and (delta, -C) on the key's line table, one shared accumulator; the subgroup check of
// B; the final exponentiation -- and k_g16_vm_verdict reads the results.  A, B and the subgroup check run side by side.  Envelopes with a point
// at infinity in the proof (valid encodings that drop a pair from the product) are counted and left to the lane-per-chain kernels.
#include <hip/hip_runtime.h>
#include <vector>
#include "g16_verify.h"
#include "fq2vm.h"
#include "g16_verify_launch.h"
using namespace zkp;

__global__ void __launch_bounds__(256) k_fq2vm(fq2vm::Launch L) {
    extern __shared__ uint32_t fq2vm_regs[];
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
    for (auto& e : T.ev) if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc = -1;
    if (rc) { g16_vm_free(T); return -1; }          // a partial upload leaves nothing behind
    T.ready = true;
    return 0;
}
void g16_vm_free(G16VmTables& T) {
    if (T.code) (void)hipFree((void*)T.code);
    if (T.off) (void)hipFree((void*)T.off);
    for (auto q : T.side) if (q) (void)hipStreamDestroy(q);
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
