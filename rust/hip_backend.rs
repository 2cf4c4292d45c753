//! The HIP backend behind libzkp's backend surface (`src/backend/mod.rs:5-8`) and `advanced::process_batch`.
//!
//! UNBUILT SOURCE (no cargo/rustc in the development image).  Drop in as `src/backend/hip.rs` next to `hip_ffi.rs`, gate
//! both behind a `hip` cargo feature, and route the call sites listed in INTEGRATION.md section 2 here.  Signatures,
//! empty-vector-on-failure and `Result<_, String>` conventions are those of the CPU backends they replace.
use super::hip_ffi as ffi;
use super::ZkpBackend;
use crate::utils::composition::BatchOperation;
use crate::utils::error_handling::{ZkpError, ZkpResult};

fn read_u64_le(data: &[u8], off: usize) -> Option<u64> {
    data.get(off..off + 8).map(|b| u64::from_le_bytes(b.try_into().unwrap()))
}

/// `impl ZkpBackend for StarkBackend` (backend/stark.rs:216-255): data = old || new, little endian.
pub struct HipStarkBackend;

impl ZkpBackend for HipStarkBackend {
    fn prove(data: &[u8]) -> Vec<u8> {
        if data.len() != 16 {
            return vec![];
        }
        let (old, new) = match (read_u64_le(data, 0), read_u64_le(data, 8)) {
            (Some(o), Some(n)) => (o, n),
            _ => return vec![],
        };
        // The C ABI returns the whole version-2 envelope (proof::improvement_proof::prove_improvement); the trait method
        // returns the bare STARK bytes (stark.rs:151-186): strip header(10) + old/new(16) and the 32-byte commitment.
        match prove_improvement_envelope(old, new) {
            Ok(env) if env.len() > 58 => env[26..env.len() - 32].to_vec(),
            _ => vec![],
        }
    }

    fn verify(proof: &[u8], data: &[u8]) -> bool {
        if data.len() != 16 {
            return false;
        }
        let (old, new) = match (read_u64_le(data, 0), read_u64_le(data, 8)) {
            (Some(o), Some(n)) => (o, n),
            _ => return false,
        };
        // rebuild the envelope verify_improvement expects (improvement_proof.rs:28-34, commitment.rs:38-50)
        let mut payload = Vec::with_capacity(16 + proof.len());
        payload.extend_from_slice(&old.to_le_bytes());
        payload.extend_from_slice(&new.to_le_bytes());
        payload.extend_from_slice(proof);
        let commitment = match crate::utils::commitment::commit_improvement(old, new) {
            Ok(c) => c,
            Err(_) => return false,
        };
        let env = crate::proof::Proof::new(5, payload, commitment).to_bytes();
        let (lens, olds, mut ok) = ([env.len() as u32], [old], [0u8]);
        let rc = unsafe { ffi::zkp_hip_verify_improvement_batch(1, env.as_ptr(), env.len() as u64, lens.as_ptr(), olds.as_ptr(), ok.as_mut_ptr()) };
        rc == 0 && ok[0] == 1
    }
}

/// `SnarkBackend::verify` is the other trait entry the proof layer reaches (proof_helpers.rs:186-206): data = the
/// public inputs; this backend verifies whole envelopes, so callers use `verify_equality_envelope` below instead.
pub struct HipSnarkBackend;

fn one<F>(stride: usize, f: F) -> Result<Vec<u8>, (i32, String)>
where
    F: FnOnce(*mut u8, u64, *mut u32, *mut i32) -> i32,
{
    let mut out = vec![0u8; stride];
    let (mut len, mut st) = ([0u32], [0i32]);
    let rc = f(out.as_mut_ptr(), stride as u64, len.as_mut_ptr(), st.as_mut_ptr());
    if rc < 0 {
        return Err((ffi::ZKP_HIP_BACKEND_ERROR, ffi::last_error()));
    }
    if st[0] != 0 {
        return Err((st[0], format!("operation failed with status {}", st[0])));
    }
    out.truncate(len[0] as usize);
    Ok(out)
}

fn to_zkp_error(e: (i32, String)) -> ZkpError {
    match e.0 {
        ffi::ZKP_HIP_INVALID_INPUT => ZkpError::InvalidInput(e.1),
        ffi::ZKP_HIP_PROOF_GENERATION_FAILED => ZkpError::ProofGenerationFailed(e.1),
        ffi::ZKP_HIP_INVALID_PROOF_FORMAT => ZkpError::InvalidProofFormat(e.1),
        _ => ZkpError::BackendError(e.1),
    }
}

/// proof::range_proof::prove_range_with_bits (range_proof.rs:16-27): the whole envelope, validation included.
pub fn prove_range_envelope(value: u64, min: u64, max: u64, n_bits: u32) -> ZkpResult<Vec<u8>> {
    crate::utils::validation::validate_range_params(value, min, max)?; // keeps the reference's messages
    let stride = unsafe { ffi::zkp_hip_range_proof_bytes(n_bits) } as usize;
    if stride == 0 {
        return Err(ZkpError::BackendError(format!("unsupported bit width {}", n_bits)));
    }
    one(stride, |o, s, l, st| unsafe {
        ffi::zkp_hip_prove_range_batch(1, &value, &min, &max, n_bits, std::ptr::null(), o, s, l, st)
    })
    .map_err(to_zkp_error)
}

/// proof::threshold_proof::prove_threshold_with_bits (threshold_proof.rs:17-32)
pub fn prove_threshold_envelope(values: &[u64], threshold: u64, n_bits: u32) -> ZkpResult<Vec<u8>> {
    let _sum = crate::utils::validation::validate_threshold_params(values, threshold)?;
    let stride = unsafe { ffi::zkp_hip_threshold_proof_bytes(n_bits) } as usize;
    let count = [values.len() as u32];
    one(stride, |o, s, l, st| unsafe {
        ffi::zkp_hip_prove_threshold_batch(1, values.as_ptr(), count.as_ptr(), &threshold, n_bits, std::ptr::null(), o, s, l, st)
    })
    .map_err(to_zkp_error)
}

/// proof::consistency_proof::prove_consistency (consistency_proof.rs:12-22)
pub fn prove_consistency_envelope(data: &[u64]) -> ZkpResult<Vec<u8>> {
    crate::utils::validation::validate_consistency_params(data)?;
    let stride = unsafe { ffi::zkp_hip_consistency_proof_bytes(data.len() as u32) } as usize;
    let count = [data.len() as u32];
    one(stride, |o, s, l, st| unsafe {
        ffi::zkp_hip_prove_consistency_batch(1, data.as_ptr(), count.as_ptr(), std::ptr::null(), o, s, l, st)
    })
    .map_err(to_zkp_error)
}

/// proof::equality_proof::prove_equality (equality_proof.rs:10-32)
pub fn prove_equality_envelope(val1: u64, val2: u64) -> ZkpResult<Vec<u8>> {
    crate::utils::validation::validate_equality_params(val1, val2)?;
    one(298, |o, s, l, st| unsafe { ffi::zkp_hip_prove_equality_batch(1, &val1, &val2, std::ptr::null(), o, s, l, st) }).map_err(to_zkp_error)
}

/// proof::set_membership::prove_membership (set_membership.rs:12-38)
pub fn prove_membership_envelope(value: u64, set: &[u64]) -> ZkpResult<Vec<u8>> {
    crate::utils::validation::validate_membership_params(value, set)?;
    let count = [set.len() as u32];
    one(10 + 4 + 8 * set.len() + 256 + 32, |o, s, l, st| unsafe {
        ffi::zkp_hip_prove_membership_batch(1, &value, set.as_ptr(), count.as_ptr(), std::ptr::null(), o, s, l, st)
    })
    .map_err(to_zkp_error)
}

/// proof::improvement_proof::prove_improvement (improvement_proof.rs:10-35)
pub fn prove_improvement_envelope(old: u64, new: u64) -> ZkpResult<Vec<u8>> {
    let _diff = crate::utils::validation::validate_improvement_params(old, new)?;
    let stride = unsafe { ffi::zkp_hip_improvement_max_bytes() } as usize;
    one(stride, |o, s, l, st| unsafe { ffi::zkp_hip_prove_improvement_batch(1, &old, &new, o, s, l, st) }).map_err(to_zkp_error)
}

/// SnarkBackend key files (snark.rs:31-38,72-139): hand the reference's own `{prefix}_pk.bin` bytes to every GPU.
pub fn load_proving_key(kind: i32, pk_bytes: &[u8]) -> ZkpResult<()> {
    let rc = unsafe { ffi::zkp_hip_groth16_load_key(kind, pk_bytes.as_ptr(), pk_bytes.len() as u64) };
    if rc != 0 {
        return Err(ZkpError::ConfigError(ffi::last_error()));
    }
    Ok(())
}

/// One process drives every GPU of the node: call once at start-up (advanced::process_batch then shards each batch).
pub fn init_all_gpus(n_gpus: u32) -> ZkpResult<()> {
    let devs: Vec<i32> = (0..n_gpus as i32).collect();
    let rc = unsafe { ffi::zkp_hip_init_devices(n_gpus, devs.as_ptr()) };
    if rc != 0 {
        return Err(ZkpError::ConfigError(ffi::last_error()));
    }
    Ok(())
}

/// Replacement of the rayon map in advanced::process_batch (advanced/batch.rs:123-131): one FFI call for the whole
/// batch; order of results = order of `batch_add_*`; any failed op fails the batch (collect::<ZkpResult<_>>).
pub fn process_batch_operations(ops: &[BatchOperation]) -> ZkpResult<Vec<Vec<u8>>> {
    let mut lists: Vec<u64> = Vec::new();
    let mut list = |v: &Vec<u64>| {
        let off = lists.len() as u64;
        lists.extend_from_slice(v);
        (v.len() as u32, off)
    };
    let raw: Vec<ffi::zkp_hip_op> = ops
        .iter()
        .map(|op| match op {
            BatchOperation::RangeProof { value, min, max } => ffi::zkp_hip_op { kind: ffi::OP_RANGE, count: 0, a: *value, b: *min, c: *max, list_off: 0 },
            BatchOperation::EqualityProof { val1, val2 } => ffi::zkp_hip_op { kind: ffi::OP_EQUALITY, count: 0, a: *val1, b: *val2, c: 0, list_off: 0 },
            BatchOperation::ThresholdProof { values, threshold } => {
                let (n, o) = list(values);
                ffi::zkp_hip_op { kind: ffi::OP_THRESHOLD, count: n, a: *threshold, b: 0, c: 0, list_off: o }
            }
            BatchOperation::MembershipProof { value, set } => {
                let (n, o) = list(set);
                ffi::zkp_hip_op { kind: ffi::OP_MEMBERSHIP, count: n, a: *value, b: 0, c: 0, list_off: o }
            }
            BatchOperation::ImprovementProof { old, new } => ffi::zkp_hip_op { kind: ffi::OP_IMPROVEMENT, count: 0, a: *old, b: *new, c: 0, list_off: 0 },
            BatchOperation::ConsistencyProof { data } => {
                let (n, o) = list(data);
                ffi::zkp_hip_op { kind: ffi::OP_CONSISTENCY, count: n, a: 0, b: 0, c: 0, list_off: o }
            }
        })
        .collect();
    let n = raw.len();
    if n == 0 {
        return Ok(vec![]);
    }
    if lists.is_empty() {
        lists.push(0); // never dereferenced, keeps the pointer non-null
    }
    let mut cap = 0u64;
    if unsafe { ffi::zkp_hip_process_batch_bytes(n as u64, raw.as_ptr(), &mut cap) } != 0 {
        return Err(ZkpError::BackendError(ffi::last_error()));
    }
    let (mut off, mut st) = (vec![0u64; n + 1], vec![0i32; n]);
    let mut out = vec![0u8; cap as usize];
    // seeds = NULL: fresh OS randomness per op, the reference's behaviour (bulletproofs.rs:82-87, snark.rs:363)
    let rc = unsafe {
        ffi::zkp_hip_process_batch(n as u64, raw.as_ptr(), lists.as_ptr(), std::ptr::null(), out.as_mut_ptr(), cap, off.as_mut_ptr(), st.as_mut_ptr())
    };
    if rc < 0 {
        return Err(ZkpError::BackendError(ffi::last_error()));
    }
    if rc > 0 {
        let i = st.iter().position(|&s| s != 0).unwrap_or(0);
        return Err(to_zkp_error((st[i], format!("batch operation {} failed with status {}", i, st[i]))));
    }
    Ok((0..n).map(|i| out[off[i] as usize..off[i + 1] as usize].to_vec()).collect())
}
