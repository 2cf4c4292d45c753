/* ORACLE (test infrastructure only; never linked into or called by the product path).
 * CPU restatement of libzkp's Bulletproofs-backed prove/verify path.  PARITY UNPINNED at the proof-byte
 * level (reference not buildable here, no byte vectors in its tests, randomised proofs) -- see
 * oracle/py/bulletproofs.py for what is and is not pinned.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load the library built from this directory.
 *
 * Every function cites the reference lines it follows in zkp_oracle.c.
 */
#ifndef ZKP_ORACLE_H
#define ZKP_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { /* status codes = ZkpError variants, /root/reference/src/utils/error_handling.rs:8-18 */
    ZKP_ORACLE_OK = 0,
    ZKP_ORACLE_INVALID_INPUT = 1,
    ZKP_ORACLE_PROOF_GENERATION_FAILED = 2,
    ZKP_ORACLE_INVALID_PROOF_FORMAT = 4,
    ZKP_ORACLE_BACKEND_ERROR = 5,
    ZKP_ORACLE_BUFFER_TOO_SMALL = 100
};

void zkp_oracle_init(void);

/* RangeProof::prove_single under Transcript::new(label); proof_out = 32*(9+2*log2 n) bytes, commit_out = 32 bytes. */
int zkp_oracle_prove_single(const char* label, uint64_t v, const uint8_t blinding[32], uint32_t n_bits,
                            const uint8_t seed[32], uint32_t proof_idx, uint8_t* proof_out, uint8_t commit_out[32]);
int zkp_oracle_verify_single(const char* label, const uint8_t* proof, uint32_t proof_len, const uint8_t commit[32], uint32_t n_bits);

/* proof::range_proof::prove_range_with_bits -> envelope bytes (1478 for n_bits = 64). */
int zkp_oracle_prove_range(uint64_t value, uint64_t min, uint64_t max, uint32_t n_bits, const uint8_t seed[32],
                           uint8_t* out, uint32_t cap, uint32_t* out_len);
int zkp_oracle_verify_range(const uint8_t* proof, uint32_t len, uint64_t min, uint64_t max);
int zkp_oracle_prove_threshold(const uint64_t* values, uint32_t count, uint64_t threshold, uint32_t n_bits,
                               const uint8_t seed[32], uint8_t* out, uint32_t cap, uint32_t* out_len);
int zkp_oracle_verify_threshold(const uint8_t* proof, uint32_t len, uint64_t threshold);
int zkp_oracle_prove_consistency(const uint64_t* data, uint32_t count, const uint8_t seed[32], uint8_t* out, uint32_t cap, uint32_t* out_len);
int zkp_oracle_verify_consistency(const uint8_t* proof, uint32_t len);

/* batch forms (OpenMP over independent ops, like rayon in /root/reference/src/advanced/batch.rs:123-131) */
int zkp_oracle_prove_range_batch(uint64_t n, const uint64_t* value, const uint64_t* min, const uint64_t* max, uint32_t n_bits,
                                 const uint8_t* seeds, uint8_t* out, uint64_t stride, uint32_t* out_len, int32_t* status, int nthreads);
int zkp_oracle_verify_range_batch(uint64_t n, const uint8_t* proofs, uint64_t stride, const uint32_t* len, const uint64_t* min,
                                  const uint64_t* max, uint8_t* ok, int nthreads);

/* ---- SNARK path (groth16.c; PARITY UNPINNED, pinned to oracle/py/groth16.py): kind 0 = equality_mimc, 1 = membership_mimc;
 * pk = ark-serialize uncompressed ProvingKey<Bn254>, the reference's own {prefix}_pk.bin format (snark.rs:31-38) */
int zkp_oracle_g16_load_key(int kind, const uint8_t* pk, uint64_t len);
int zkp_oracle_snark_commit_value(uint64_t value, uint8_t out[32]);
int zkp_oracle_prove_equality(uint64_t val1, uint64_t val2, const uint8_t seed[32], uint8_t* out, uint32_t cap, uint32_t* out_len);
int zkp_oracle_prove_membership(uint64_t value, const uint64_t* set, uint32_t count, const uint8_t seed[32], uint8_t* out, uint32_t cap, uint32_t* out_len);
/* ---- STARK path (stark.c; PARITY UNPINNED, pinned to oracle/py/stark.py): deterministic, cap >= 3527 */
int zkp_oracle_prove_improvement(uint64_t old_value, uint64_t new_value, uint8_t* out, uint32_t cap, uint32_t* out_len);

/* advanced::process_batch (/root/reference/src/advanced/batch.rs:110-140,262-283): one OpenMP task per op, like the rayon
 * par_iter; same op record and output convention as zkp_hip_process_batch (proof i at out[out_off[i] .. out_off[i+1])).
 * Returns 1 if any op failed, ZKP_ORACLE_BUFFER_TOO_SMALL if out_cap is too small (out_off[n] = needed). */
typedef struct zkp_oracle_op { uint32_t kind, count; uint64_t a, b, c, list_off; } zkp_oracle_op;
int zkp_oracle_process_batch(uint64_t n, const zkp_oracle_op* ops, const uint64_t* lists, const uint8_t* seeds,
                             uint8_t* out, uint64_t out_cap, uint64_t* out_off, int32_t* status, int nthreads);

/* tape + generator access for kernel-level parity tests */
void zkp_oracle_tape_draw64(const uint8_t seed[32], uint32_t proof_idx, uint32_t slot, uint8_t out[64]);
void zkp_oracle_generator(uint32_t index, uint8_t enc[32]); /* 0 = B, 1 = B_blinding, 2+i = G_i, 66+i = H_i (party 0) */
/* instrumented operation counts of the calling thread since the last reset */
void zkp_oracle_counters(uint64_t* fe_mul, uint64_t* sc_mul, int reset);
#ifdef __cplusplus
}
#endif
#endif
