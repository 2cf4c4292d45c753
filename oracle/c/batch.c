/* ORACLE (test infrastructure only; never linked into or called by the product path).
 * advanced::process_batch (/root/reference/src/advanced/batch.rs:110-140): the reference maps process_batch_operation
 * (batch.rs:262-283) over the ops with rayon and collects into Result<Vec<Vec<u8>>>; here one OpenMP task per op.  This is
 * what bench.py's cpu_baseline times for the mixed batch (kind "port": the arithmetic underneath is this directory's C
 * restatement of upstream's algorithms, not the dalek / arkworks / winterfell crates). */
#include "zkp_oracle.h"
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static uint32_t op_cap(const zkp_oracle_op* o) {
    switch (o->kind) {
        case 1: return 1478;
        case 2: return 298;
        case 3: return 762;
        case 4: return 10 + 4 + 8 * (o->count <= 64 ? o->count : 0) + 256 + 32;
        case 5: return 3527;
        case 6: return o->count ? 10 + 4 + 32 * o->count + (4 + 672 + 32) * (o->count - 1) + 32 : 0;
        default: return 0;
    }
}

int zkp_oracle_process_batch(uint64_t n, const zkp_oracle_op* ops, const uint64_t* lists, const uint8_t* seeds,
                             uint8_t* out, uint64_t out_cap, uint64_t* out_off, int32_t* status, int nthreads) {
    zkp_oracle_init();
    uint64_t* slot = (uint64_t*)malloc(8 * (n + 1));
    uint64_t tot = 0;
    for (uint64_t i = 0; i < n; i++) { slot[i] = tot; tot += op_cap(&ops[i]); }
    slot[n] = tot;
    uint8_t* tmp = (uint8_t*)malloc(tot ? tot : 1);
    uint32_t* len = (uint32_t*)calloc(n ? n : 1, 4);
    int fail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(dynamic, 1) reduction(| : fail)
#endif
    for (uint64_t i = 0; i < n; i++) {
        const zkp_oracle_op* o = &ops[i];
        const uint8_t* sd = seeds + 32 * i;
        uint8_t* dst = tmp + slot[i];
        const uint32_t cap = (uint32_t)(slot[i + 1] - slot[i]);
        uint32_t l = 0; int rc;
        switch (o->kind) {                                       /* batch.rs:262-283 */
            case 1: rc = zkp_oracle_prove_range(o->a, o->b, o->c, 64, sd, dst, cap, &l); break;
            case 2: rc = zkp_oracle_prove_equality(o->a, o->b, sd, dst, cap, &l); break;
            case 3: rc = zkp_oracle_prove_threshold(lists + o->list_off, o->count, o->a, 64, sd, dst, cap, &l); break;
            case 4: rc = o->count <= 64 ? zkp_oracle_prove_membership(o->a, lists + o->list_off, o->count, sd, dst, cap, &l) : ZKP_ORACLE_INVALID_INPUT; break;
            case 5: rc = zkp_oracle_prove_improvement(o->a, o->b, dst, cap, &l); break;
            case 6: rc = o->count ? zkp_oracle_prove_consistency(lists + o->list_off, o->count, sd, dst, cap, &l) : ZKP_ORACLE_INVALID_INPUT; break;
            default: rc = ZKP_ORACLE_INVALID_INPUT; break;
        }
        status[i] = rc; len[i] = rc == 0 ? l : 0; fail |= rc != 0;
    }
    (void)nthreads;
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; i++) { out_off[i] = total; total += len[i]; }
    out_off[n] = total;
    int ret = fail;
    if (total > out_cap) ret = ZKP_ORACLE_BUFFER_TOO_SMALL;
    else for (uint64_t i = 0; i < n; i++) memcpy(out + out_off[i], tmp + slot[i], len[i]);
    free(slot); free(tmp); free(len);
    return ret;
}
