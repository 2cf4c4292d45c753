// Fq2 virtual machine of the Groth16 verifier (the check is ark-groth16's pairing product under
// /root/reference/src/backend/snark.rs:377-401,455-495).  The Miller loop, the G2 subgroup check and the final exponentiation of an
// envelope are fixed straight-line programs over Fq2 (tools/gen_fq2vm.py writes them as micro-operation tables, fq2vm_programs.h); this
// header is the interpreter.  One workgroup = K wavefronts x 32 envelopes: lanes l and l + 32 of every wave work for envelope l, the 2 K
// half-waves split the operations of a round between them, and the Fq2 register file of the 32 envelopes lives in LDS
// ([register][word][envelope]: every access of a half-wave is to consecutive banks).  A barrier closes each round.  Why: the lane-per-chain kernels (g16_verify.h) were
// a lane's serial chain through ~500 KB of straight-line code per Miller iteration (instruction-cache misses on every line) at 64-256
// waves on 1024 SIMDs; here the whole interpreter is a few thousand instructions and a chain's products run on the four SIMDs of a CU.
#pragma once
#include <vector>
#include <utility>
#include "bn254_pairing.h"
#include "fq2vm_programs.h"

namespace zkp { namespace fq2vm {

enum Op : uint32_t { NOP, MUL, SQ, ADD, SUB, MULXI, CONJ, MUL0, MUL1, INV, LDG, STG, LDC, MOV, NEG, LDK, STC, T3M, T3P, END };
constexpr uint32_t BAR = 0x80u, FQ2_W = 20;

// 3a - 2b and 3a + 2b (the output step of a cyclotomic squaring) as one operation each: limb-wise (3a < 12p with limbs < 2^28; 2b safe for the
// 8p-borrowing subtraction: limbs < 2^27, value < 8p), one weak reduction per coordinate
ZKP_HD inline fq fq_t3(const fq& a, const fq& b, bool minus) {
    const fq a3 = fq_add_l(fq_dbl_l(a), a), b2 = fq_dbl_l(b);
    return fq_reduce_weak(minus ? fq_sub_k8(a3, b2) : fq_add_l(a3, b2));
}
ZKP_HD inline fq2 vm_t3(const fq2& a, const fq2& b, bool minus) { return fq2{fq_t3(a.c0, b.c0, minus), fq_t3(a.c1, b.c1, minus)}; }
// arithmetic operations: inputs and outputs "safe" in the vocabulary of bn254_fq.h (carried, value < 4p)
ZKP_HD inline fq2 alu(uint32_t op, const fq2& a, const fq2& b) {
    switch (op) {
    case MUL: return f_mul(a, b);
    case SQ: return f_sq(a);
    case ADD: return f_add(a, b);
    case SUB: return f_sub(a, b);
    case MULXI: return fq2_mul_xi(a);
    case CONJ: return fq2_conj(a);
    case MUL0: return fq2_mul_fq(a, b.c0);
    case MUL1: return fq2_mul_fq(a, b.c1);
    case INV: return f_inv(a);
    case NEG: return f_neg(a);
    case T3M: return vm_t3(a, b, true);
    case T3P: return vm_t3(a, b, false);
    default: return a;          // MOV
    }
}
ZKP_HD inline bool op_has_b(uint32_t op) { return op == MUL || op == ADD || op == SUB || op == MUL0 || op == MUL1 || op == T3M || op == T3P; }
ZKP_HD inline void fq2_to_words(uint32_t w[FQ2_W], const fq2& a) { for (int k = 0; k < 10; k++) { w[k] = a.c0.v[k]; w[10 + k] = a.c1.v[k]; } }
ZKP_HD inline fq2 fq2_from_words(const uint32_t w[FQ2_W]) { fq2 a; for (int k = 0; k < 10; k++) { a.c0.v[k] = w[k]; a.c1.v[k] = w[10 + k]; } return a; }

// tables of one K (device or host pointers)
struct Tables { const uint32_t* code; const uint32_t* off; uint32_t K; const uint32_t* consts; };
// one launch: `script` (program | cursor advance << 8) runs for envelopes [0, n) on the slot buffer io[slot][20][n]; kconst = the launch's
// constant table, read by LDK at cursor + operand (the final exponentiation's key constant; chain B's line table)
struct Launch { Tables T; const uint16_t* script; uint32_t script_len; uint32_t n; uint32_t* io; uint64_t chain_stride; const uint32_t* kconst; };

// A micro-operation is two words: op | barrier << 7 | dst << 8 | a << 16 | b << 24 for lanes 0-31 of the wave and
// present | dst << 8 | a << 16 | b << 24 for lanes 32-63: the two halves of a wave run the same operation on different registers of the
// same 32 envelopes (G = 32 envelopes per workgroup; all 64 lanes do useful work).
constexpr uint32_t G = 32;

// host emulation: one envelope, the K streams of every round one after the other (tests/emul; also documents the semantics)
inline void run_host(const Launch& L, uint32_t i, uint32_t nregs) {
    std::vector<fq2> reg(nregs);
    for (auto& r : reg) { f_set_zero(r); }
    auto io_at = [&](uint32_t slot, uint32_t w) -> uint32_t& { return L.io[((size_t)slot * FQ2_W + w) * L.n + i]; };
    uint32_t cur = 0;
    for (uint32_t s = 0; s < L.script_len; cur += L.script[s] >> 8, s++) {
        std::vector<uint32_t> pc(L.T.K);
        for (uint32_t w = 0; w < L.T.K; w++) pc[w] = L.T.off[(size_t)(L.script[s] & 255u) * L.T.K + w];
        bool more = true;
        while (more) {
            std::vector<std::pair<uint32_t, fq2>> writes;          // a round's results land together, as on the device after the barrier
            for (uint32_t w = 0; w < L.T.K; w++) {
                std::vector<std::pair<uint32_t, fq2>> local;
                auto get = [&](uint32_t r) { for (auto it = local.rbegin(); it != local.rend(); ++it) if (it->first == r) return it->second; return reg[r]; };
                for (;;) {
                    const uint32_t u0 = L.T.code[pc[w]], u1 = L.T.code[pc[w] + 1];
                    const uint32_t op = u0 & 0x7fu;
                    if (op == END) { more = false; break; }
                    pc[w] += 2;
                    std::vector<std::pair<uint32_t, fq2>> step;      // both halves read before either writes
                    for (int h = 0; h < 2; h++) {
                        if (h == 1 && !(u1 & 1u)) continue;
                        const uint32_t t = h ? u1 : u0, d = (t >> 8) & 255u, a = (t >> 16) & 255u, b = t >> 24;
                        uint32_t wd[FQ2_W];
                        if (op == LDG) { for (uint32_t k = 0; k < FQ2_W; k++) wd[k] = io_at(b, k); step.emplace_back(d, fq2_from_words(wd)); }
                        else if (op == LDC) step.emplace_back(d, fq2_from_words(L.T.consts + (size_t)b * FQ2_W));
                        else if (op == LDK) step.emplace_back(d, fq2_from_words(L.kconst + (size_t)(cur + b) * FQ2_W));
                        else if (op == STG || op == STC) { fq2_to_words(wd, get(a)); for (uint32_t k = 0; k < FQ2_W; k++) io_at(b + (op == STC ? cur : 0u), k) = wd[k]; }
                        else if (op != NOP) step.emplace_back(d, alu(op, get(a), op_has_b(op) ? get(b) : get(a)));
                    }
                    for (auto& e : step) local.push_back(e);
                    if (u0 & BAR) break;
                }
                for (auto& e : local) writes.push_back(e);
            }
            for (auto& e : writes) reg[e.first] = e.second;
        }
    }
}

#if defined(__HIPCC__)
// device interpreter; dynamic LDS = nregs * 20 * G words, [register][word][envelope]
__device__ inline void run_device(const Launch& L, uint32_t* __restrict__ regs) {
    const uint32_t lane = threadIdx.x & 63u, gl = lane & (G - 1u);
    const bool hi = lane >= G;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t i = blockIdx.x * G + gl;
    const bool inb = i < L.n;
    uint32_t* const io = L.io + (size_t)blockIdx.y * L.chain_stride;
    const uint32_t* __restrict__ code = L.T.code;
    auto ld = [&](uint32_t r) { fq2 x; const uint32_t* q = regs + r * (FQ2_W * G) + gl; ZKP_UNROLL for (uint32_t k = 0; k < 10; k++) { x.c0.v[k] = q[k * G]; x.c1.v[k] = q[(10 + k) * G]; } return x; };
    auto st = [&](bool on, uint32_t r, const fq2& x) { if (on) { uint32_t* q = regs + r * (FQ2_W * G) + gl; ZKP_UNROLL for (uint32_t k = 0; k < 10; k++) { q[k * G] = x.c0.v[k]; q[(10 + k) * G] = x.c1.v[k]; } } };
    uint32_t cur = 0;
    for (uint32_t s = 0; s < L.script_len; s++) {
        const uint32_t ent = __builtin_amdgcn_readfirstlane((uint32_t)L.script[s]);
        uint32_t pc = __builtin_amdgcn_readfirstlane(L.T.off[(ent & 255u) * L.T.K + wave]);
        uint32_t u0 = __builtin_amdgcn_readfirstlane(code[pc]), u1 = __builtin_amdgcn_readfirstlane(code[pc + 1]);
        for (;;) {
            const uint32_t op = u0 & 0x7fu;
            if (op == END) break;
            const uint32_t t = hi ? u1 : u0;
            const bool on = !hi || (u1 & 1u);           // this half has an operation in this step (an idle half recomputes register 0 and stores nothing)
            const uint32_t d = (t >> 8) & 255u, a = (t >> 16) & 255u, b = t >> 24;
            const bool bar = u0 & BAR;
            pc += 2;
            u0 = __builtin_amdgcn_readfirstlane(code[pc]); u1 = __builtin_amdgcn_readfirstlane(code[pc + 1]);      // the next micro-operation, fetched while this one runs
            switch (op) {
            case MUL: st(on, d, f_mul(ld(a), ld(b))); break;
            case SQ: st(on, d, f_sq(ld(a))); break;
            case ADD: st(on, d, f_add(ld(a), ld(b))); break;
            case SUB: st(on, d, f_sub(ld(a), ld(b))); break;
            case MULXI: st(on, d, fq2_mul_xi(ld(a))); break;
            case CONJ: st(on, d, fq2_conj(ld(a))); break;
            case MUL0: case MUL1: { const fq2 y = ld(b); st(on, d, fq2_mul_fq(ld(a), op == MUL0 ? y.c0 : y.c1)); break; }
            case INV: st(on, d, f_inv(ld(a))); break;
            case NEG: st(on, d, f_neg(ld(a))); break;
            case T3M: case T3P: st(on, d, vm_t3(ld(a), ld(b), op == T3M)); break;
            case MOV: st(on, d, ld(a)); break;
            // loads from memory: all twenty words are requested before the first is stored (one memory latency per operation, not twenty)
            case LDG: { uint32_t t[FQ2_W]; ZKP_UNROLL for (uint32_t k = 0; k < FQ2_W; k++) t[k] = inb ? io[((size_t)b * FQ2_W + k) * L.n + i] : 0u;
                        if (on) { ZKP_UNROLL for (uint32_t k = 0; k < FQ2_W; k++) regs[(d * FQ2_W + k) * G + gl] = t[k]; } break; }
            case STG: case STC: if (on && inb) { const size_t sl = b + (op == STC ? cur : 0u); ZKP_UNROLL for (uint32_t k = 0; k < FQ2_W; k++) io[(sl * FQ2_W + k) * L.n + i] = regs[(a * FQ2_W + k) * G + gl]; } break;
            case LDC: case LDK: { const uint32_t* src = op == LDC ? L.T.consts + b * FQ2_W : L.kconst + (size_t)(cur + b) * FQ2_W;
                        uint32_t t[FQ2_W]; ZKP_UNROLL for (uint32_t k = 0; k < FQ2_W; k++) t[k] = src[k];
                        if (on) { ZKP_UNROLL for (uint32_t k = 0; k < FQ2_W; k++) regs[(d * FQ2_W + k) * G + gl] = t[k]; } break; }
            default: break;
            }
            if (bar) __syncthreads();
        }
        cur += ent >> 8;
    }
}
#endif

} }  // namespace zkp::fq2vm
