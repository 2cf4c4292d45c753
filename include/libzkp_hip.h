/* libzkp_hip -- C ABI of the MI355X (gfx950) proving backend for libzkp's Bulletproofs hot path.
 *
 * Drop-in boundary: these entry points are what the reference's Rust side binds through `extern "C"`
 * in place of its CPU backend calls (binding sketch: INTEGRATION.md).  Plain pointers and sizes only.
 * All calls are batch-first; a single proof is the n = 1 case.  Thread-safe: one context ("shard") per
 * initialised GPU, calls are serialised per shard; the batch calls drive every shard from one process.
 *
 * Randomness: the reference draws blindings from OsRng/thread_rng (bulletproofs.rs:82-87,132), so its
 * proof bytes are not reproducible.  Here every op takes a 32-byte seed from which all of its random
 * scalars are derived (tape definition: DESIGN.md); pass seeds = NULL to have the library draw fresh
 * seeds from the OS RNG (the reference's behaviour).
 *
 * Status codes per item follow ZkpError (/root/reference/src/utils/error_handling.rs:8-18).
 */
#ifndef LIBZKP_HIP_H
#define LIBZKP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ZKP_HIP_RANGE_PROOF_BYTES 1478u      /* proof::range_proof::prove_range output, n_bits = 64 */
#define ZKP_HIP_THRESHOLD_PROOF_BYTES 762u   /* proof::threshold_proof::prove_threshold output      */

enum {
    ZKP_HIP_OK = 0,
    ZKP_HIP_INVALID_INPUT = 1,             /* ZkpError::InvalidInput          */
    ZKP_HIP_PROOF_GENERATION_FAILED = 2,   /* ZkpError::ProofGenerationFailed */
    ZKP_HIP_INVALID_PROOF_FORMAT = 4,      /* ZkpError::InvalidProofFormat    */
    ZKP_HIP_BACKEND_ERROR = 5              /* ZkpError::BackendError          */
};

/* Library-level return values: 0 = every item succeeded, 1 = at least one item has status != 0
 * (outputs of the other items are still valid), < 0 = the call itself failed (see zkp_hip_last_error). */
/* No entry point lets a C++ exception out (the reference turns every failure into an Err: batch.rs:126-130): a failed host allocation or any
 * other exception inside the library comes back as ZKP_HIP_E_RUNTIME with a message, and the library stays usable.  One call accepts at most
 * 2^22 operations and 2^28 list values; beyond that ZKP_HIP_E_ARGUMENT, before anything is read or sized from the arguments. */
#define ZKP_HIP_E_RUNTIME (-1)       /* HIP runtime error / no device / out of (host or device) memory / C++ exception */
#define ZKP_HIP_E_UNSUPPORTED (-2)   /* e.g. n_bits other than 8, 16, 32, 64 */
#define ZKP_HIP_E_ARGUMENT (-3)

/* One-time setup on HIP device `device`: derives the 130 Bulletproofs generators (PedersenGens::default,
 * BulletproofGens::new party 0; replaces bp_gens_pair_bits, bulletproofs.rs:61-80), builds the
 * fixed-base window tables on the device (radix 2^16: 8.7 GB of HBM per GPU, self-checked slot against slot; plus 208 MB of radix-1024
 * tables for the verifier).  Idempotent.  Called implicitly (device 0) by the prove calls.
 * Every device initialised this way becomes one SHARD of the library (numbered in registration order). */
int zkp_hip_init(int device);
/* Multi-GPU (replaces the rayon fan-out of batch.rs:123-131 at node scale): registers `count` shards, shard k on
 * HIP device devices[k], each initialised by its own host thread.  One process then drives all of them:
 * zkp_hip_process_batch / zkp_hip_batch_* cut every variant's ops into one contiguous slice per shard, and
 * zkp_hip_groth16_load_key / _generate_key install the ONE trusted setup on every shard.  The same HIP device may be listed
 * twice (two independent contexts; the one-GPU tests run the multi-shard path this way).  Idempotent for an identical list;
 * any other re-registration needs zkp_hip_shutdown first. */
int zkp_hip_init_devices(uint32_t count, const int* devices);
int zkp_hip_device_count(void);                      /* shards registered */
/* The per-variant entry points (everything except the batch calls below) run on ONE shard: shard 0 unless the calling
 * thread selected another one here (thread-local; concurrent callers on different shards do not serialise). */
int zkp_hip_use_device(int shard);
/* Releases every device resource of every shard (tables, loaded keys, workspaces, pooled staging buffers, streams, pinned
 * staging) and forgets the shard registration; a later call initialises again (keys must be loaded again; batches staged
 * earlier must be staged again).  Registered with atexit() at the first initialisation, so the HIP runtime never finds live
 * objects of this library during its own exit-time teardown. */
void zkp_hip_shutdown(void);
/* Thread-local description of the last failure of a call made on this thread. */
const char* zkp_hip_last_error(void);

/* Replaces a loop of proof::range_proof::prove_range_with_bits(value, min, max, n_bits) -- prove_range is n_bits = 64 --
 * (/root/reference/src/proof/range_proof.rs:10-27 -> BulletproofsBackend::prove_range_with_bounds_bits,
 * /root/reference/src/backend/bulletproofs.rs:112-178) as issued by process_batch
 * (/root/reference/src/advanced/batch.rs:123-131,264-266).
 *   value,min,max : n host u64 each           seeds : 32*n host bytes or NULL
 *   n_bits        : 8, 16, 32 or 64, one width per call (others: ZKP_HIP_E_UNSUPPORTED, upstream's InvalidBitsize)
 *   out           : n records of `stride` bytes; record i holds the version-2 Proof envelope of
 *                   zkp_hip_range_proof_bytes(n_bits) bytes (1478 for 64 bits, 128 less per halving)
 *   out_len[i]    : that size on success, 0 on failure      status[i] : per-item code (validation.rs:5-18; a width
 *                   value - min or max - value does not fit in is ZKP_HIP_INVALID_INPUT, bulletproofs.rs:121-129) */
uint64_t zkp_hip_range_proof_bytes(uint32_t n_bits);        /* 0 for an unsupported width */
int zkp_hip_prove_range_batch(uint64_t n, const uint64_t* value, const uint64_t* min, const uint64_t* max, uint32_t n_bits,
                              const uint8_t* seeds, uint8_t* out, uint64_t stride, uint32_t* out_len, int32_t* status);

/* Same contract with every pointer a DEVICE pointer (inputs already resident in HBM, proofs left in HBM);
 * `stream` is a hipStream_t (NULL = the library's own stream).  Asynchronous with respect to the host
 * unless `any_failed` (host int, may be NULL) is requested, in which case the call synchronises the stream. */
int zkp_hip_prove_range_batch_device(uint64_t n, const uint64_t* d_value, const uint64_t* d_min, const uint64_t* d_max, uint32_t n_bits,
                                     const uint8_t* d_seeds, uint8_t* d_out, uint64_t stride, uint32_t* d_out_len, int32_t* d_status,
                                     void* stream, int* any_failed);

/* Replaces a loop of proof::threshold_proof::prove_threshold_with_bits(values, threshold, n_bits) -- prove_threshold is
 * n_bits = 64 -- (/root/reference/src/proof/threshold_proof.rs:12-32 -> bulletproofs.rs:309-366).  values = all ops' value
 * lists concatenated, counts[i] = length of op i's list.  One envelope (scheme 3) of
 * zkp_hip_threshold_proof_bytes(n_bits) bytes per op (762 for 64 bits, 64 less per halving); stride >= that. */
uint64_t zkp_hip_threshold_proof_bytes(uint32_t n_bits);    /* 0 for an unsupported width */
int zkp_hip_prove_threshold_batch(uint64_t n, const uint64_t* values, const uint32_t* counts, const uint64_t* thresholds, uint32_t n_bits,
                                  const uint8_t* seeds, uint8_t* out, uint64_t stride, uint32_t* out_len, int32_t* status);

/* Replaces a loop of proof::consistency_proof::prove_consistency(data)
 * (/root/reference/src/proof/consistency_proof.rs:12-22 -> bulletproofs.rs:368-437).  data = all ops' lists concatenated.
 * Envelope (scheme 6) size depends on the list length: zkp_hip_consistency_proof_bytes(count); stride >= the largest. */
uint64_t zkp_hip_consistency_proof_bytes(uint32_t count);
int zkp_hip_prove_consistency_batch(uint64_t n, const uint64_t* data, const uint32_t* counts, const uint8_t* seeds,
                                    uint8_t* out, uint64_t stride, uint32_t* out_len, int32_t* status);

/* ---- Groth16 / BN254 (equality and set-membership circuits, /root/reference/src/backend/snark.rs) ----
 * kind: 0 = EqualityCircuit ("equality_mimc"), 1 = MembershipCircuit ("membership_mimc", 64 slots).
 * pk = ark-serialize *uncompressed* ProvingKey<Bn254> bytes, i.e. the content of the reference's
 * `{prefix}_pk.bin` key files (snark.rs:31-38,97-112).  Builds the fixed-base tables of every key point on the GPU. */
int zkp_hip_groth16_load_key(int kind, const uint8_t* pk, uint64_t len);
/* What the loaded key of `kind` holds on the calling thread's shard: the radix of its fixed-base window tables (2^*wbits; *uneven = 1:
 * the 18-window form of radix 2^14) and the HBM they occupy (shared by the shards of one GPU).  Default policy: radix 2^13, ~34 GB for
 * the two circuits together; ZKP_HIP_G16_TABLE_BUDGET_MB=<MB per key> opts into larger tables (2^14-uneven, ~72 GB, measured 1.7 %
 * faster on the mixed batch), ZKP_HIP_G16_WBITS=8..15 forces a radix; a device with less free memory gets a smaller radix.  The
 * reference keeps a ProvingKey in host memory (snark.rs:40-56); this is the device-side cost of its replacement.  Any pointer may be NULL. */
int zkp_hip_groth16_key_info(int kind, uint32_t* wbits, uint32_t* uneven, uint64_t* table_bytes);

/* Circuit-specific trusted setup (replaces Groth16::circuit_specific_setup at snark.rs:318,337): toxic waste from
 * `setup_seed` (32 bytes; NULL = OS randomness, the reference's behaviour), every key point computed on the GPU.
 * The key is loaded into the backend and also returned in ark-serialize uncompressed form so the caller can persist
 * `{prefix}_pk.bin` / `{prefix}_vk.bin` exactly like load_or_generate_setup (snark.rs:122-139).  Call with NULL buffers
 * first to learn the sizes (the key is generated and loaded either way). */
int zkp_hip_groth16_generate_key(int kind, const uint8_t* setup_seed, uint8_t* pk_out, uint64_t pk_cap, uint64_t* pk_len,
                                 uint8_t* vk_out, uint64_t vk_cap, uint64_t* vk_len);

/* utils::commitment::commit_value_snark (commitment.rs:14-16; MiMC-5/110 over BN254 Fr, snark.rs:201-221): 32 bytes per value. */
int zkp_hip_snark_commit_value_batch(uint64_t n, const uint64_t* values, uint8_t* out);

/* Replaces a loop of proof::equality_proof::prove_equality(val1, val2) (equality_proof.rs:10-32 ->
 * SnarkBackend::prove_equality_zk, snark.rs:343-374).  One 298-byte envelope (scheme 2) per op; stride >= 298. */
int zkp_hip_prove_equality_batch(uint64_t n, const uint64_t* val1, const uint64_t* val2, const uint8_t* seeds,
                                 uint8_t* out, uint64_t stride, uint32_t* out_len, int32_t* status);

/* Replaces a loop of proof::range_proof::verify_range(proof, min, max) (range_proof.rs:28-47 ->
 * BulletproofsBackend::verify_range_with_bounds, bulletproofs.rs:181-295).  proofs = n envelopes at `stride` bytes,
 * lens[i] bytes used.  ok[i] = 1 accepted, 0 rejected (malformed framing, wrong bounds, invalid points/scalars, failed
 * verification equation).  The bit width (8, 16, 32 or 64) is read from each envelope (bulletproofs.rs:211-216), so one
 * batch may mix widths.  The two verification equations of RangeProof::verify_single are folded with a transcript-derived
 * weight (as upstream does with a random one). */
int zkp_hip_verify_range_batch(uint64_t n, const uint8_t* proofs, uint64_t stride, const uint32_t* lens,
                               const uint64_t* mins, const uint64_t* maxs, uint8_t* ok);
/* Same for proof::threshold_proof::verify_threshold(proof, threshold) (threshold_proof.rs:34-47 -> bulletproofs.rs:550-626):
 * one RangeProof per envelope (scheme 3) whose commitment must be C - threshold*B. */
int zkp_hip_verify_threshold_batch(uint64_t n, const uint8_t* proofs, uint64_t stride, const uint32_t* lens,
                                   const uint64_t* thresholds, uint8_t* ok);
/* Same for proof::consistency_proof::verify_consistency(proof) (consistency_proof.rs:24-32 -> bulletproofs.rs:439-547):
 * k commitments whose SHA-256 is the envelope commitment, k - 1 RangeProofs of the successive differences. */
int zkp_hip_verify_consistency_batch(uint64_t n, const uint8_t* proofs, uint64_t stride, const uint32_t* lens, uint8_t* ok);

/* Replaces a loop of proof::improvement_proof::prove_improvement(old, new) (improvement_proof.rs:10-35 ->
 * StarkBackend::prove / prove_improvement, stark.rs:151-186,216-235; commitment utils/commitment.rs:38-50).
 * Deterministic (no randomness tape).  Envelope (scheme 5) = 10 + 16 + stark + 32 bytes, where the STARK proof's length
 * depends on how many of the 32 query positions coincide; stride >= zkp_hip_improvement_max_bytes() (3527).
 * new <= old -> ZKP_HIP_INVALID_INPUT ("new value must be greater than old value", validation.rs:63-71). */
uint32_t zkp_hip_improvement_max_bytes(void);
int zkp_hip_prove_improvement_batch(uint64_t n, const uint64_t* old_values, const uint64_t* new_values,
                                    uint8_t* out, uint64_t stride, uint32_t* out_len, int32_t* status);
/* Same with every pointer a device pointer; launched on `stream` (NULL = the library's stream, synchronised before
 * return).  Ops with new <= old get out_len = 0. */
int zkp_hip_prove_improvement_batch_device(uint64_t n, const uint64_t* d_old, const uint64_t* d_new, uint8_t* d_out, uint64_t stride,
                                           uint32_t* d_out_len, void* stream);

/* Groth16 verification of equality (scheme 2, 298 B) / membership (scheme 4) envelopes under the verifying key that leads
 * the loaded proving key (zkp_hip_groth16_load_key / _generate_key): SnarkBackend::verify_equality_zk (snark.rs:377-401) and
 * verify_membership_zk (snark.rs:455-495) as reached from verify_proof_cryptographic (proof_helpers.rs:180-206).  The public
 * inputs are the envelope's own commitment (and embedded set); callers compare those with what they expect, as
 * equality_proof.rs:34-60 / set_membership.rs:40-70 do.  ok[i] = 1 accepted / 0 rejected.
 * Calls with more than 8192 envelopes first try ONE weighted pairing check for all of them (fresh 128-bit weights from getrandom;
 * libzkp_amd/csrc/g16_rlc.h) and verify envelope by envelope only if it does not stand, so every verdict is the per-envelope one up to a
 * soundness error of 2^-128 per call; ZKP_HIP_NO_BATCH_VERIFY=1 / ZKP_HIP_G16_BATCH_VERIFY_MIN move or remove the threshold. */
int zkp_hip_verify_equality_batch(uint64_t n, const uint8_t* proofs, uint64_t stride, const uint32_t* lens, uint8_t* ok);
int zkp_hip_verify_membership_batch(uint64_t n, const uint8_t* proofs, uint64_t stride, const uint32_t* lens, uint8_t* ok);

/* Replaces a loop of proof::improvement_proof::verify_improvement(proof, old) (improvement_proof.rs:37-68 ->
 * StarkBackend::verify, stark.rs:190-211,237-255): ok[i] = 1 accepted / 0 rejected (framing, stored old != old, binding
 * commitment, transcript-derived checks, Merkle openings, DEEP / remainder consistency). */
int zkp_hip_verify_improvement_batch(uint64_t n, const uint8_t* proofs, uint64_t stride, const uint32_t* lens,
                                     const uint64_t* old_values, uint8_t* ok);

/* Replaces a loop of proof::set_membership::prove_membership(value, set) (set_membership.rs:12-38 ->
 * SnarkBackend::prove_membership_zk, snark.rs:405-452).  sets = all ops' sets concatenated, set_counts[i] <= 64.
 * Envelope (scheme 4) = 10 + 4 + 8*len + 256 + 32 bytes; stride >= the largest. */
int zkp_hip_prove_membership_batch(uint64_t n, const uint64_t* values, const uint64_t* sets, const uint32_t* set_counts, const uint8_t* seeds,
                                   uint8_t* out, uint64_t stride, uint32_t* out_len, int32_t* status);

/* Replaces advanced::process_batch (/root/reference/src/advanced/batch.rs:110-140,262-283): a mixed list of independent
 * operations, one batched device call per variant, proofs returned in the caller's order.  `kind` = the envelope
 * scheme id of the operation (proof/mod.rs).  Fields by kind:
 *   RANGE (value a, min b, max c) | EQUALITY (val1 a, val2 b) | IMPROVEMENT (old a, new b)
 *   THRESHOLD (threshold a, `count` values at lists[list_off..]) | MEMBERSHIP (value a, set of `count` <= 64 values)
 *   CONSISTENCY (`count` values at lists[list_off..]).
 * seeds: 32 bytes per op (NULL = OS randomness, the reference's behaviour; ignored by IMPROVEMENT, which is deterministic).
 * Proof i occupies out[out_off[i] .. out_off[i+1]); out_off has n+1 entries and out_off[n] is the total (also when
 * out_cap is too small: then -3 is returned and nothing is written).  status[i] as for the per-variant calls; returns 1
 * if any op failed (the reference fails the whole batch, batch.rs:126-130: callers should then discard the output). */
enum { ZKP_HIP_OP_RANGE = 1, ZKP_HIP_OP_EQUALITY = 2, ZKP_HIP_OP_THRESHOLD = 3, ZKP_HIP_OP_MEMBERSHIP = 4, ZKP_HIP_OP_IMPROVEMENT = 5, ZKP_HIP_OP_CONSISTENCY = 6 };
typedef struct zkp_hip_op {
    uint32_t kind;
    uint32_t count;
    uint64_t a, b, c;
    uint64_t list_off;
} zkp_hip_op;
int zkp_hip_process_batch(uint64_t n, const zkp_hip_op* ops, const uint64_t* lists, const uint8_t* seeds,
                          uint8_t* out, uint64_t out_cap, uint64_t* out_off, int32_t* status);
/* Capacity that is enough for any outcome of zkp_hip_process_batch on these ops (improvement envelopes counted at
 * zkp_hip_improvement_max_bytes()): callers size `out` with it instead of proving twice.  No device work. */
int zkp_hip_process_batch_bytes(uint64_t n, const zkp_hip_op* ops, uint64_t* max_total);

/* How the batch calls spread ops over GPUs: shard_of_op[i] = the shard that proves op i when `shards` shards are
 * registered.  Every variant's ops, in the caller's order, are cut into `shards` contiguous slices of equal size (+-1), so all
 * GPUs run the same kernel mix.  Pure host logic: needs no device. */
int zkp_hip_plan_shards(uint64_t n, const zkp_hip_op* ops, uint32_t shards, uint32_t* shard_of_op);

/* The three phases of zkp_hip_process_batch as separate calls.  A staged batch is the device-side counterpart of the
 * reference's ProofBatch (composition.rs:337-413, filled by batch_add_*, batch.rs:40-108): ops bucketed by variant, validated
 * (validation.rs), cut into per-shard slices and uploaded, i.e. resident in HBM.  zkp_hip_batch_prove runs the whole batch on
 * every shard (variants on their own streams, proofs packed in op order on the device) and waits; it may be repeated.
 * zkp_hip_batch_fetch copies the proofs out exactly as zkp_hip_process_batch returns them (same return values). */
typedef struct zkp_hip_batch zkp_hip_batch;
int zkp_hip_batch_stage(uint64_t n, const zkp_hip_op* ops, const uint64_t* lists, const uint8_t* seeds, zkp_hip_batch** batch);
int zkp_hip_batch_prove(zkp_hip_batch* batch);
/* The same in two halves: _async enqueues the whole proving on every shard and returns; _wait blocks until it is done (fetch and
 * device_results wait by themselves).  A shard keeps two lanes of streams and workspaces, so a caller that stages batch k + 1 and
 * launches it before waiting for batch k has two batches in flight: the latency-bound tail of one runs under the MSMs of the next. */
int zkp_hip_batch_prove_async(zkp_hip_batch* batch);
int zkp_hip_batch_wait(zkp_hip_batch* batch);
uint64_t zkp_hip_batch_max_bytes(const zkp_hip_batch* batch);
int zkp_hip_batch_fetch(zkp_hip_batch* batch, uint8_t* out, uint64_t out_cap, uint64_t* out_off, int32_t* status);
/* Device-resident results of a proved batch, shard by shard (a multi-process driver gathers them with RCCL instead of going
 * through the host): copies shard `shard`'s packed proofs -- its ops in ascending op order -- into d_out (device memory, cap
 * >= zkp_hip_batch_max_bytes) and its n_ops + 1 byte offsets into d_out_off (device, may be NULL), ordered on `stream` (a
 * hipStream_t; NULL = the shard's own stream, synchronised before return). */
int zkp_hip_batch_device_results(zkp_hip_batch* batch, uint32_t shard, uint8_t* d_out, uint64_t cap, uint64_t* d_out_off, uint64_t* n_ops, void* stream);
void zkp_hip_batch_free(zkp_hip_batch* batch);

/* Kernel timing for the roofline line of bench.py: when enabled, every launch of the dominant kernel
 * (fixed-base MSM) is bracketed by hipEvents on its own stream. */
void zkp_hip_profile_enable(int on);
/* Synchronises, then returns accumulated MSM kernel time (ms), launch count, and table-entry gathers
 * (point additions) since the last reset, summed over the shards.  zkp_hip_profile_read is kernel 0. */
enum { ZKP_HIP_KERNEL_MSM_ED25519 = 0, ZKP_HIP_KERNEL_MSM_BN254_G1 = 1, ZKP_HIP_KERNEL_MSM_BN254_G2 = 2 };
int zkp_hip_profile_read_kernel(int which, double* ms, uint64_t* launches, uint64_t* point_adds, int reset);
int zkp_hip_profile_read(double* msm_ms, uint64_t* msm_launches, uint64_t* msm_point_adds, int reset);
/* Tunables (benchmarking).  window budget: 0 = chunking chosen per launch from the batch size (default); 32*T = slot-aligned
 * chunks of 32*T windows; 10000 + c = the window-granular layout with about c chunks.  sub-batches: independent slices
 * of a range batch on separate HIP streams (default 1: measured slower; takes effect at the next
 * zkp_hip_init).  msm variant: values >= 100 scale the resident-workgroup count the chunk choice aims at (x100). */
void zkp_hip_set_window_budget(uint32_t budget);
void zkp_hip_set_subbatches(uint32_t n);
void zkp_hip_set_msm_variant(uint32_t v);

#ifdef __cplusplus
}
#endif
#endif
