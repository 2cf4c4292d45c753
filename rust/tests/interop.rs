//! Pins libzkp_hip to the REAL libzkp, in both directions, and dumps the result as golden vectors.
//!
//! UNBUILT SOURCE (the development image has no cargo / rustc; nothing in this file has been compiled or run).  For a maintainer who
//! has both: copy to `tests/hip_interop.rs` of a libzkp checkout that carries `src/backend/hip_ffi.rs` (this repository's
//! `rust/hip_ffi.rs`, made `pub mod hip_ffi` behind the `hip` feature), then
//!
//!     LIBZKP_HIP_DIR=<libzkp-amd>/libzkp_amd/lib LIBZKP_HIP_VECTORS=<libzkp-amd>/tests/golden/reference \
//!         cargo test --no-default-features --features hip --test hip_interop -- --nocapture --test-threads 1
//!
//! on a machine with an MI355X.  What it checks (every `assert!` is a parity claim this repository cannot make by itself):
//!   1. the HIP backend's envelopes of all six schemes are accepted by the reference's `verify_*` (tests/integration.rs:14-57 shapes);
//!   2. the reference's envelopes are accepted by the HIP verifiers;
//!   3. a proving key written by the reference (`LIBZKP_SNARK_KEY_DIR`, snark.rs:31-38,122-139) loads through
//!      `zkp_hip_groth16_load_key`, and proofs made with it on the GPU verify under the reference -- i.e. the R1CS matrices, variable
//!      order and QAP reduction are ark-groth16's;
//!   4. `prove_improvement` -- the one deterministic scheme -- is BYTE-IDENTICAL (stark.rs:151-186);
//!   5. `commit_value_snark` (MiMC constants, snark.rs:186-221) is byte-identical;
//!   6. (round 4) the form of the device tables is invisible in the bytes: the reference's key loaded under every radix policy this
//!      library has (default 2^13, the 2^14 opt-in in its even and its 18-window form, a forced small radix) gives the SAME envelope for
//!      the same seed, and the reference accepts it -- the radix-2^16 generator tables of the range prover and the scratch-free STARK
//!      kernel are covered by claims 1 and 4, whose envelopes they now produce;
//!   7. (round 4) Groth16 envelopes with a point at infinity -- crafted from a reference envelope, the encodings ark-serialize accepts --
//!      get the SAME verdict from the reference and from the HIP verifiers (they leave the Fq2 machine for the lane-per-chain kernels).
//!   8. (round 4) one pairing check per batch (libzkp_amd/csrc/g16_rlc.h): 8200 reference-made equality envelopes in ONE call -- above the
//!      library's threshold of 8193 -- are all accepted; with one of them put under another's commitment the call names exactly that
//!      envelope, as the reference's `verify_equality` does for it.  (Run the whole test once more with `ZKP_HIP_G16_BATCH_VERIFY_MIN=1` --
//!      read once per process -- to send the single-envelope calls of claims 2 and 7 through the batch check as well.)  The 16-lane
//!      public-input accumulation and the verifier's HBM generator tables are what claims 2 and 7 run on now.
//! and what it writes into $LIBZKP_HIP_VECTORS (consumed by tests/test_reference_vectors.py, which skips while the directory is empty):
//!   reference_envelopes.json   envelopes made by the reference            -> our verifiers must keep accepting them
//!   hip_envelopes.json         inputs + seeds + envelopes made on the GPU that the reference accepted -> our prover must keep producing exactly them
//!   improvement_vectors.json   (old, new, envelope) from the reference    -> our prover must produce exactly them
//!   snark_commitments.json     (value, commitment) from the reference
//!   special_envelopes.json     crafted envelopes (points at infinity) with the reference's verdict -> our verifiers must give the same
//!   equality_mimc_{pk,vk}.bin, membership_mimc_{pk,vk}.bin               the reference's own trusted setup
use libzkp::backend::hip_ffi as ffi;
use libzkp::proof::{consistency_proof, equality_proof, improvement_proof, range_proof, set_membership, threshold_proof};
use libzkp::utils::commitment::commit_value_snark;
use std::fmt::Write as _;
use std::path::PathBuf;

fn hex(b: &[u8]) -> String {
    let mut s = String::with_capacity(2 * b.len());
    for x in b {
        write!(s, "{:02x}", x).unwrap();
    }
    s
}
fn out_dir() -> PathBuf {
    PathBuf::from(std::env::var("LIBZKP_HIP_VECTORS").expect("LIBZKP_HIP_VECTORS names the output directory"))
}
fn seed(i: u64) -> [u8; 32] {
    let mut s = [0u8; 32];
    s[..8].copy_from_slice(&i.to_le_bytes());
    s[8] = 0xA5;
    s
}
fn list_json(v: &[u64]) -> String {
    format!("[{}]", v.iter().map(|x| x.to_string()).collect::<Vec<_>>().join(", "))
}

/// one op through zkp_hip_process_batch with a fixed seed; returns the envelope
fn hip_prove(op: ffi::zkp_hip_op, lists: &[u64], sd: &[u8; 32]) -> Vec<u8> {
    let mut cap = 0u64;
    assert_eq!(unsafe { ffi::zkp_hip_process_batch_bytes(1, &op, &mut cap) }, 0);
    let mut out = vec![0u8; cap as usize];
    let (mut off, mut st) = ([0u64; 2], [0i32; 1]);
    let lp = if lists.is_empty() { std::ptr::null() } else { lists.as_ptr() };
    let rc = unsafe { ffi::zkp_hip_process_batch(1, &op, lp, sd.as_ptr(), out.as_mut_ptr(), cap, off.as_mut_ptr(), st.as_mut_ptr()) };
    assert_eq!((rc, st[0]), (0, 0), "{}", ffi::last_error());
    out.truncate(off[1] as usize);
    out
}
fn op(kind: u32, count: u32, a: u64, b: u64, c: u64) -> ffi::zkp_hip_op {
    ffi::zkp_hip_op { kind, count, a, b, c, list_off: 0 }
}

#[test]
fn hip_backend_and_reference_accept_each_other() {
    let dir = out_dir();
    std::fs::create_dir_all(&dir).unwrap();
    // the reference's own setup, persisted where both sides can read it
    std::env::set_var("LIBZKP_SNARK_KEY_DIR", &dir);
    let ref_eq = equality_proof::prove_equality(42, 42).expect("reference prove_equality (generates and persists the key)");
    let ref_mem = set_membership::prove_membership(2, vec![1, 2, 3]).expect("reference prove_membership");
    assert_eq!(unsafe { ffi::zkp_hip_init(0) }, 0, "{}", ffi::last_error());
    for (kind, name) in [(0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")] {
        let pk = std::fs::read(dir.join(name)).expect("the reference persisted its proving key");
        assert_eq!(unsafe { ffi::zkp_hip_groth16_load_key(kind, pk.as_ptr(), pk.len() as u64) }, 0, "{}", ffi::last_error());        // claim 3a
    }

    // ---- 2. reference envelopes -> HIP verifiers
    let mut refs = String::from("[\n");
    let ref_range = range_proof::prove_range(7, 0, 10).unwrap();
    let ref_thr = threshold_proof::prove_threshold(vec![3, 4, 5], 10).unwrap();
    let ref_con = consistency_proof::prove_consistency(vec![1, 2, 3]).unwrap();
    let ref_imp = improvement_proof::prove_improvement(1, 5).unwrap();
    let mut ok = [0u8; 1];
    let l = |e: &Vec<u8>| [e.len() as u32];
    unsafe {
        assert_eq!(ffi::zkp_hip_verify_range_batch(1, ref_range.as_ptr(), ref_range.len() as u64, l(&ref_range).as_ptr(), [0u64].as_ptr(), [10u64].as_ptr(), ok.as_mut_ptr()), 0);
        assert_eq!(ok[0], 1, "reference range envelope rejected by the HIP verifier");
        assert_eq!(ffi::zkp_hip_verify_threshold_batch(1, ref_thr.as_ptr(), ref_thr.len() as u64, l(&ref_thr).as_ptr(), [10u64].as_ptr(), ok.as_mut_ptr()), 0);
        assert_eq!(ok[0], 1, "reference threshold envelope rejected");
        assert_eq!(ffi::zkp_hip_verify_consistency_batch(1, ref_con.as_ptr(), ref_con.len() as u64, l(&ref_con).as_ptr(), ok.as_mut_ptr()), 0);
        assert_eq!(ok[0], 1, "reference consistency envelope rejected");
        assert_eq!(ffi::zkp_hip_verify_equality_batch(1, ref_eq.as_ptr(), ref_eq.len() as u64, l(&ref_eq).as_ptr(), ok.as_mut_ptr()), 0);
        assert_eq!(ok[0], 1, "reference equality envelope rejected");
        assert_eq!(ffi::zkp_hip_verify_membership_batch(1, ref_mem.as_ptr(), ref_mem.len() as u64, l(&ref_mem).as_ptr(), ok.as_mut_ptr()), 0);
        assert_eq!(ok[0], 1, "reference membership envelope rejected");
        assert_eq!(ffi::zkp_hip_verify_improvement_batch(1, ref_imp.as_ptr(), ref_imp.len() as u64, l(&ref_imp).as_ptr(), [1u64].as_ptr(), ok.as_mut_ptr()), 0);
        assert_eq!(ok[0], 1, "reference improvement envelope rejected");
    }
    for (scheme, args, e) in [("range", "{\"min\": 0, \"max\": 10}", &ref_range), ("threshold", "{\"threshold\": 10}", &ref_thr), ("consistency", "{}", &ref_con),
                              ("equality", "{\"value\": 42}", &ref_eq), ("membership", "{\"set\": [1, 2, 3]}", &ref_mem), ("improvement", "{\"old\": 1}", &ref_imp)] {
        writeln!(refs, "  {{\"scheme\": \"{}\", \"verify_args\": {}, \"envelope\": \"{}\"}},", scheme, args, hex(e)).unwrap();
    }
    refs.truncate(refs.len() - 2);
    refs.push_str("\n]\n");
    std::fs::write(dir.join("reference_envelopes.json"), refs).unwrap();

    // ---- 1 / 3b. HIP envelopes -> reference verifiers
    let mut hips = String::from("[\n");
    let mut n = 0u64;
    let mut emit = |scheme: &str, fields: String, sd: &[u8; 32], e: &[u8]| {
        writeln!(hips, "  {{\"scheme\": \"{}\", {}, \"seed\": \"{}\", \"envelope\": \"{}\"}},", scheme, fields, hex(sd), hex(e)).unwrap();
    };
    for (v, lo, hi) in [(7u64, 0u64, 10u64), (0, 0, 0), (1 << 32, 0, 1 << 32), (u64::MAX, u64::MAX - 5, u64::MAX), (50, 0, 100)] {
        n += 1; let sd = seed(n);
        let e = hip_prove(op(ffi::OP_RANGE, 0, v, lo, hi), &[], &sd);
        assert!(range_proof::verify_range(e.clone(), lo, hi), "HIP range envelope rejected by the reference: prove_range({}, {}, {})", v, lo, hi);
        emit("range", format!("\"value\": {}, \"min\": {}, \"max\": {}", v, lo, hi), &sd, &e);
    }
    for (vals, t) in [(vec![3u64, 4, 5], 10u64), (vec![1 << 40], 0), (vec![5, 5], 10)] {
        n += 1; let sd = seed(n);
        let e = hip_prove(op(ffi::OP_THRESHOLD, vals.len() as u32, t, 0, 0), &vals, &sd);
        assert!(threshold_proof::verify_threshold(e.clone(), t), "HIP threshold envelope rejected by the reference");
        emit("threshold", format!("\"values\": {}, \"threshold\": {}", list_json(&vals), t), &sd, &e);
    }
    for vals in [vec![1u64, 2, 3], vec![9], vec![0, 0, 1 << 50, u64::MAX]] {
        n += 1; let sd = seed(n);
        let e = hip_prove(op(ffi::OP_CONSISTENCY, vals.len() as u32, 0, 0, 0), &vals, &sd);
        assert!(consistency_proof::verify_consistency(e.clone()), "HIP consistency envelope rejected by the reference");
        emit("consistency", format!("\"values\": {}", list_json(&vals)), &sd, &e);
    }
    for v in [3u64, 42, 0, u64::MAX] {
        n += 1; let sd = seed(n);
        let e = hip_prove(op(ffi::OP_EQUALITY, 0, v, v, 0), &[], &sd);
        assert!(equality_proof::verify_equality(e.clone(), v, v), "HIP equality envelope rejected by the reference (reference's key, GPU prover)");        // claim 3b
        assert!(equality_proof::verify_equality_with_commitment(e.clone(), commit_value_snark(v)));
        emit("equality", format!("\"value\": {}", v), &sd, &e);
    }
    for (v, set) in [(2u64, vec![1u64, 2, 3]), (7, vec![7]), (63, (0..64).collect::<Vec<u64>>())] {
        n += 1; let sd = seed(n);
        let e = hip_prove(op(ffi::OP_MEMBERSHIP, set.len() as u32, v, 0, 0), &set, &sd);
        assert!(set_membership::verify_membership(e.clone(), set.clone()), "HIP membership envelope rejected by the reference");
        emit("membership", format!("\"value\": {}, \"set\": {}", v, list_json(&set)), &sd, &e);
    }
    hips.truncate(hips.len() - 2);
    hips.push_str("\n]\n");
    std::fs::write(dir.join("hip_envelopes.json"), hips).unwrap();

    // ---- 4. the deterministic scheme: byte parity
    let mut imps = String::from("[\n");
    for (old, new) in [(1u64, 5u64), (30, 50), (0, 1), (5, 1 << 63), (u64::MAX - 1, u64::MAX), (1 << 62, (1 << 62) + (1 << 31) + 7)] {
        let want = improvement_proof::prove_improvement(old, new).unwrap();
        let got = hip_prove(op(ffi::OP_IMPROVEMENT, 0, old, new, 0), &[], &seed(0));
        assert_eq!(got, want, "prove_improvement({}, {}) differs from the reference's bytes", old, new);
        assert!(improvement_proof::verify_improvement(got, old));
        writeln!(imps, "  {{\"old\": {}, \"new\": {}, \"envelope\": \"{}\"}},", old, new, hex(&want)).unwrap();
    }
    imps.truncate(imps.len() - 2);
    imps.push_str("\n]\n");
    std::fs::write(dir.join("improvement_vectors.json"), imps).unwrap();

    // ---- 5. MiMC commitments
    let mut cms = String::from("[\n");
    for v in [0u64, 1, 42, 1 << 32, u64::MAX] {
        let want = commit_value_snark(v);
        let mut got = [0u8; 32];
        assert_eq!(unsafe { ffi::zkp_hip_snark_commit_value_batch(1, &v, got.as_mut_ptr()) }, 0);
        assert_eq!(&got[..], &want[..], "snark_commit_value({}) differs", v);
        writeln!(cms, "  {{\"value\": {}, \"commitment\": \"{}\"}},", v, hex(&want)).unwrap();
    }
    cms.truncate(cms.len() - 2);
    cms.push_str("\n]\n");
    std::fs::write(dir.join("snark_commitments.json"), cms).unwrap();

    // ---- 6. table forms: same key, same seed, same bytes whatever the radix policy of the device tables
    {
        let sd = seed(777);
        let eq_default = hip_prove(op(ffi::OP_EQUALITY, 0, 42, 42, 0), &[], &sd);
        let mem_default = hip_prove(op(ffi::OP_MEMBERSHIP, 3, 2, 0, 0), &[1, 2, 3], &sd);
        for (name, value) in [("ZKP_HIP_G16_TABLE_BUDGET_MB", "60000"), ("ZKP_HIP_G16_UNEVEN", "0"), ("ZKP_HIP_G16_WBITS", "11")] {
            std::env::set_var(name, value);
            if name == "ZKP_HIP_G16_UNEVEN" { std::env::set_var("ZKP_HIP_G16_TABLE_BUDGET_MB", "60000"); }
            for (kind, file) in [(0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")] {
                let pk = std::fs::read(dir.join(file)).unwrap();
                assert_eq!(unsafe { ffi::zkp_hip_groth16_load_key(kind, pk.as_ptr(), pk.len() as u64) }, 0, "{}", ffi::last_error());
            }
            let (mut wb, mut un, mut bytes) = (0u32, 0u32, 0u64);
            assert_eq!(unsafe { ffi::zkp_hip_groth16_key_info(0, &mut wb, &mut un, &mut bytes) }, 0);
            eprintln!("table form under {}={}: radix 2^{} uneven {} ({} MB)", name, value, wb, un, bytes >> 20);
            assert_eq!(hip_prove(op(ffi::OP_EQUALITY, 0, 42, 42, 0), &[], &sd), eq_default, "equality envelope depends on the table form ({}={})", name, value);
            assert_eq!(hip_prove(op(ffi::OP_MEMBERSHIP, 3, 2, 0, 0), &[1, 2, 3], &sd), mem_default, "membership envelope depends on the table form");
            std::env::remove_var("ZKP_HIP_G16_TABLE_BUDGET_MB"); std::env::remove_var("ZKP_HIP_G16_UNEVEN"); std::env::remove_var("ZKP_HIP_G16_WBITS");
        }
        for (kind, file) in [(0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")] {          // back to the default policy
            let pk = std::fs::read(dir.join(file)).unwrap();
            assert_eq!(unsafe { ffi::zkp_hip_groth16_load_key(kind, pk.as_ptr(), pk.len() as u64) }, 0);
        }
        assert!(equality_proof::verify_equality(eq_default, 42, 42));
    }

    // ---- 7. crafted points at infinity: the verdicts must agree, whatever they are (ark-serialize: flag 0x40 in the last byte, all else zero)
    {
        let mut specials = String::from("[\n");
        let base = ref_eq.clone();                                  // [10-byte header][A 64][B 128][C 64][commitment 32]
        for (what, at, len) in [("A", 10usize, 64usize), ("B", 74, 128), ("C", 202, 64)] {
            let mut e = base.clone();
            for b in &mut e[at..at + len] { *b = 0; }
            e[at + len - 1] = 0x40;
            let want = equality_proof::verify_equality(e.clone(), 42, 42);
            assert_eq!(unsafe { ffi::zkp_hip_verify_equality_batch(1, e.as_ptr(), e.len() as u64, [e.len() as u32].as_ptr(), ok.as_mut_ptr()) }, 0);
            assert_eq!(ok[0] == 1, want, "equality envelope with {} at infinity: reference says {}, HIP says {}", what, want, ok[0]);
            writeln!(specials, "  {{\"scheme\": \"equality\", \"what\": \"{} at infinity\", \"verify_args\": {{\"value\": 42}}, \"envelope\": \"{}\", \"reference_verdict\": {}}},", what, hex(&e), want).unwrap();
        }
        specials.truncate(specials.len() - 2);
        specials.push_str("\n]\n");
        std::fs::write(dir.join("special_envelopes.json"), specials).unwrap();
    }

    // ---- 8. one pairing check per batch: many reference envelopes in one call; the verdicts are the reference's, one by one
    {
        let n = 8200usize;                                           // above the default threshold (8193): the batch check runs first
        let mut envs: Vec<Vec<u8>> = Vec::with_capacity(64);
        for v in 0..64u64 { envs.push(equality_proof::prove_equality(1000 + v, 1000 + v).expect("reference prove_equality")); }
        let width = envs[0].len();
        let mut buf = vec![0u8; n * width];
        for j in 0..n { buf[j * width..(j + 1) * width].copy_from_slice(&envs[j % 64]); }
        let lens = vec![width as u32; n];
        let mut oks = vec![0u8; n];
        assert_eq!(unsafe { ffi::zkp_hip_verify_equality_batch(n as u64, buf.as_ptr(), width as u64, lens.as_ptr(), oks.as_mut_ptr()) }, 0, "{}", ffi::last_error());
        assert!(oks.iter().all(|&x| x == 1), "a reference-made equality envelope was rejected inside a large batch");
        let (bad, donor) = (5000usize, 5001usize);                   // envelope 5000 under its neighbour's commitment: every point still valid
        let tail: Vec<u8> = buf[donor * width + 266..donor * width + 298].to_vec();
        buf[bad * width + 266..bad * width + 298].copy_from_slice(&tail);
        assert_eq!(unsafe { ffi::zkp_hip_verify_equality_batch(n as u64, buf.as_ptr(), width as u64, lens.as_ptr(), oks.as_mut_ptr()) }, 0);
        let crossed = buf[bad * width..(bad + 1) * width].to_vec();
        assert!(!equality_proof::verify_equality(crossed, 1000 + (bad % 64) as u64, 1000 + (bad % 64) as u64));
        for j in 0..n { assert_eq!(oks[j] == 1, j != bad, "envelope {} of the large batch: HIP verdict {}", j, oks[j]); }
    }

    // negative cases of tests/integration.rs:72-91 against HIP envelopes
    let mut bad = hip_prove(op(ffi::OP_RANGE, 0, 7, 0, 10), &[], &seed(99));
    bad[12] ^= 0xFF;
    assert!(!range_proof::verify_range(bad, 0, 10));
    let e = hip_prove(op(ffi::OP_EQUALITY, 0, 3, 3, 0), &[], &seed(98));
    assert!(!equality_proof::verify_equality(e, 3, 4));
    unsafe { ffi::zkp_hip_shutdown() };
}
