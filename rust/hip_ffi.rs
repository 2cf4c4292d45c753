//! `extern "C"` bindings of libzkp_hip (include/libzkp_hip.h), one declaration per exported symbol.
//!
//! UNBUILT SOURCE: the image this backend was developed in has no cargo/rustc, so this file has never been compiled.
//! It is what a libzkp maintainer adds as `src/backend/hip_ffi.rs` (feature `hip`); the same symbols are exercised
//! end to end through the C ABI by `tests/abi_call_all.cpp` (C++) and `libzkp_amd/_native.py` (ctypes).
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_int, c_void};

pub const ZKP_HIP_OK: i32 = 0;
pub const ZKP_HIP_INVALID_INPUT: i32 = 1; // ZkpError::InvalidInput
pub const ZKP_HIP_PROOF_GENERATION_FAILED: i32 = 2; // ZkpError::ProofGenerationFailed
pub const ZKP_HIP_INVALID_PROOF_FORMAT: i32 = 4; // ZkpError::InvalidProofFormat
pub const ZKP_HIP_BACKEND_ERROR: i32 = 5; // ZkpError::BackendError
pub const ZKP_HIP_E_RUNTIME: c_int = -1;
pub const ZKP_HIP_E_UNSUPPORTED: c_int = -2;
pub const ZKP_HIP_E_ARGUMENT: c_int = -3;

pub const OP_RANGE: u32 = 1;
pub const OP_EQUALITY: u32 = 2;
pub const OP_THRESHOLD: u32 = 3;
pub const OP_MEMBERSHIP: u32 = 4;
pub const OP_IMPROVEMENT: u32 = 5;
pub const OP_CONSISTENCY: u32 = 6;

/// `zkp_hip_op`: one BatchOperation (utils/composition.rs:343-350) flattened; `kind` = the envelope scheme id.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct zkp_hip_op {
    pub kind: u32,
    pub count: u32,
    pub a: u64,
    pub b: u64,
    pub c: u64,
    pub list_off: u64,
}

/// Opaque staged batch (`zkp_hip_batch_stage`).
#[repr(C)]
pub struct zkp_hip_batch {
    _private: [u8; 0],
}

#[link(name = "zkp_hip")]
extern "C" {
    // lifecycle / devices
    pub fn zkp_hip_init(device: c_int) -> c_int;
    pub fn zkp_hip_init_devices(count: u32, devices: *const c_int) -> c_int;
    pub fn zkp_hip_device_count() -> c_int;
    pub fn zkp_hip_use_device(shard: c_int) -> c_int;
    pub fn zkp_hip_shutdown();
    pub fn zkp_hip_last_error() -> *const c_char;

    // Bulletproofs framings (backend/bulletproofs.rs:112-178, 309-366, 368-437)
    pub fn zkp_hip_range_proof_bytes(n_bits: u32) -> u64;
    pub fn zkp_hip_threshold_proof_bytes(n_bits: u32) -> u64;
    pub fn zkp_hip_consistency_proof_bytes(count: u32) -> u64;
    pub fn zkp_hip_prove_range_batch(n: u64, value: *const u64, min: *const u64, max: *const u64, n_bits: u32, seeds: *const u8,
                                     out: *mut u8, stride: u64, out_len: *mut u32, status: *mut i32) -> c_int;
    pub fn zkp_hip_prove_range_batch_device(n: u64, d_value: *const u64, d_min: *const u64, d_max: *const u64, n_bits: u32, d_seeds: *const u8,
                                            d_out: *mut u8, stride: u64, d_out_len: *mut u32, d_status: *mut i32, stream: *mut c_void,
                                            any_failed: *mut c_int) -> c_int;
    pub fn zkp_hip_prove_threshold_batch(n: u64, values: *const u64, counts: *const u32, thresholds: *const u64, n_bits: u32, seeds: *const u8,
                                         out: *mut u8, stride: u64, out_len: *mut u32, status: *mut i32) -> c_int;
    pub fn zkp_hip_prove_consistency_batch(n: u64, data: *const u64, counts: *const u32, seeds: *const u8, out: *mut u8, stride: u64,
                                           out_len: *mut u32, status: *mut i32) -> c_int;

    // Groth16 (backend/snark.rs)
    pub fn zkp_hip_groth16_load_key(kind: c_int, pk: *const u8, len: u64) -> c_int;
    pub fn zkp_hip_groth16_key_info(kind: c_int, wbits: *mut u32, uneven: *mut u32, table_bytes: *mut u64) -> c_int;
    pub fn zkp_hip_groth16_generate_key(kind: c_int, setup_seed: *const u8, pk_out: *mut u8, pk_cap: u64, pk_len: *mut u64, vk_out: *mut u8,
                                        vk_cap: u64, vk_len: *mut u64) -> c_int;
    pub fn zkp_hip_snark_commit_value_batch(n: u64, values: *const u64, out: *mut u8) -> c_int;
    pub fn zkp_hip_prove_equality_batch(n: u64, val1: *const u64, val2: *const u64, seeds: *const u8, out: *mut u8, stride: u64,
                                        out_len: *mut u32, status: *mut i32) -> c_int;
    pub fn zkp_hip_prove_membership_batch(n: u64, values: *const u64, sets: *const u64, set_counts: *const u32, seeds: *const u8, out: *mut u8,
                                          stride: u64, out_len: *mut u32, status: *mut i32) -> c_int;

    // STARK (backend/stark.rs)
    pub fn zkp_hip_improvement_max_bytes() -> u32;
    pub fn zkp_hip_prove_improvement_batch(n: u64, old_values: *const u64, new_values: *const u64, out: *mut u8, stride: u64,
                                           out_len: *mut u32, status: *mut i32) -> c_int;
    pub fn zkp_hip_prove_improvement_batch_device(n: u64, d_old: *const u64, d_new: *const u64, d_out: *mut u8, stride: u64,
                                                  d_out_len: *mut u32, stream: *mut c_void) -> c_int;

    // verification
    pub fn zkp_hip_verify_range_batch(n: u64, proofs: *const u8, stride: u64, lens: *const u32, mins: *const u64, maxs: *const u64, ok: *mut u8) -> c_int;
    pub fn zkp_hip_verify_threshold_batch(n: u64, proofs: *const u8, stride: u64, lens: *const u32, thresholds: *const u64, ok: *mut u8) -> c_int;
    pub fn zkp_hip_verify_consistency_batch(n: u64, proofs: *const u8, stride: u64, lens: *const u32, ok: *mut u8) -> c_int;
    pub fn zkp_hip_verify_equality_batch(n: u64, proofs: *const u8, stride: u64, lens: *const u32, ok: *mut u8) -> c_int;
    pub fn zkp_hip_verify_membership_batch(n: u64, proofs: *const u8, stride: u64, lens: *const u32, ok: *mut u8) -> c_int;
    pub fn zkp_hip_verify_improvement_batch(n: u64, proofs: *const u8, stride: u64, lens: *const u32, old_values: *const u64, ok: *mut u8) -> c_int;

    // advanced::process_batch (advanced/batch.rs:110-140,262-283)
    pub fn zkp_hip_process_batch(n: u64, ops: *const zkp_hip_op, lists: *const u64, seeds: *const u8, out: *mut u8, out_cap: u64,
                                 out_off: *mut u64, status: *mut i32) -> c_int;
    pub fn zkp_hip_process_batch_bytes(n: u64, ops: *const zkp_hip_op, max_total: *mut u64) -> c_int;
    pub fn zkp_hip_plan_shards(n: u64, ops: *const zkp_hip_op, shards: u32, shard_of_op: *mut u32) -> c_int;
    pub fn zkp_hip_batch_stage(n: u64, ops: *const zkp_hip_op, lists: *const u64, seeds: *const u8, batch: *mut *mut zkp_hip_batch) -> c_int;
    pub fn zkp_hip_batch_prove(batch: *mut zkp_hip_batch) -> c_int;
    pub fn zkp_hip_batch_prove_async(batch: *mut zkp_hip_batch) -> c_int;
    pub fn zkp_hip_batch_wait(batch: *mut zkp_hip_batch) -> c_int;
    pub fn zkp_hip_batch_max_bytes(batch: *const zkp_hip_batch) -> u64;
    pub fn zkp_hip_batch_fetch(batch: *mut zkp_hip_batch, out: *mut u8, out_cap: u64, out_off: *mut u64, status: *mut i32) -> c_int;
    pub fn zkp_hip_batch_device_results(batch: *mut zkp_hip_batch, shard: u32, d_out: *mut u8, cap: u64, d_out_off: *mut u64, n_ops: *mut u64,
                                        stream: *mut c_void) -> c_int;
    pub fn zkp_hip_batch_free(batch: *mut zkp_hip_batch);

    // profiling / tunables (benchmarking)
    pub fn zkp_hip_profile_enable(on: c_int);
    pub fn zkp_hip_profile_read(msm_ms: *mut f64, msm_launches: *mut u64, msm_point_adds: *mut u64, reset: c_int) -> c_int;
    pub fn zkp_hip_profile_read_kernel(which: c_int, ms: *mut f64, launches: *mut u64, point_adds: *mut u64, reset: c_int) -> c_int;
    pub fn zkp_hip_set_window_budget(budget: u32);
    pub fn zkp_hip_set_subbatches(n: u32);
    pub fn zkp_hip_set_msm_variant(v: u32);
}

/// Thread-local text of the last failure of a call made on this thread.
pub fn last_error() -> String {
    unsafe { std::ffi::CStr::from_ptr(zkp_hip_last_error()).to_string_lossy().into_owned() }
}
