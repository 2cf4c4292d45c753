// TEST INFRASTRUCTURE: host build of the STARK device header for the no-GPU test tier.
#include "../../libzkp_amd/csrc/stark_steps.h"
#include <string.h>
using namespace zkp;
struct NoSync { void operator()() const {} };
static StarkConst g_c; static bool g_ready = false;
extern "C" {
// op 0 mul, 1 add, 2 sub, 3 neg, 4 inv
void emul_f128_op(int op, const uint64_t a[2], const uint64_t b[2], uint64_t out[2]) {
    const f128 x = f128_make(a[0], a[1]), y = f128_make(b[0], b[1]); f128 r;
    switch (op) { case 0: r = f128_mul(x, y); break; case 1: r = f128_add(x, y); break; case 2: r = f128_sub(x, y); break; case 3: r = f128_neg(x); break; default: r = f128_inv(x); }
    out[0] = r.lo; out[1] = r.hi;
}
void emul_blake3_words(const uint32_t* in, uint32_t nwords, uint32_t out[8]) { blake3_words(out, in, nwords); }
void emul_improvement_commitment(uint64_t o, uint64_t n, uint8_t out[32]) { improvement_commitment(out, o, n); }
// the whole envelope; returns its length
uint32_t emul_stark_prove(uint64_t oldv, uint64_t newv, uint8_t* out, uint32_t cap) {
    if (!g_ready) { stark_build_constants(g_c); g_ready = true; }
    static StarkMem M;
    stark_prove(M, g_c, oldv, newv, 0, 1, NoSync());
    if (M.out_len > cap) return 0;
    memcpy(out, M.out, M.out_len);
    return M.out_len;
}
uint32_t emul_stark_max_envelope() { return STARK_MAX_ENVELOPE; }
int emul_stark_verify(const uint8_t* env, uint32_t len, uint64_t oldv) {
    if (!g_ready) { stark_build_constants(g_c); g_ready = true; }
    return stark_verify_envelope(env, len, oldv, g_c) ? 1 : 0;
}
}
