"""Host-side mirror of the reference interface: validation messages, batch registry semantics (no GPU)."""
import pytest

import libzkp_amd as z


def test_validation_messages_match_reference():
    with pytest.raises(ValueError, match="^min cannot be greater than max$"):
        z.prove_range(5, 10, 3)
    with pytest.raises(ValueError, match=r"^value 50 is not in range \[0, 10\]$"):
        z.prove_range(50, 0, 10)
    with pytest.raises(OverflowError):
        z.prove_range(-1, 0, 10)
    with pytest.raises(OverflowError):
        z.prove_range(2**64, 0, 10)


def test_batch_registry_semantics():
    b = z.create_proof_batch()
    assert b != 0
    z.batch_add_range_proof(b, 5, 0, 10)
    z.batch_add_range_proof(b, 7, 0, 2**32)
    z.batch_add_threshold_proof(b, [10, 20, 30], 50)
    z.batch_add_consistency_proof(b, [1, 2, 2])
    st = z.get_batch_status(b)
    assert st == {"total_operations": 4, "range_proofs": 2, "equality_proofs": 0, "threshold_proofs": 1,
                  "membership_proofs": 0, "improvement_proofs": 0, "consistency_proofs": 1}
    with pytest.raises(ValueError, match=r"value 11 is not in range \[0, 10\]"):
        z.batch_add_range_proof(b, 11, 0, 10)                 # validated at add time (batch.rs:73-76)
    with pytest.raises(ValueError, match="values are not equal"):
        z.batch_add_equality_proof(b, 1, 2)
    with pytest.raises(ValueError, match="sum 3 is less than threshold 5"):
        z.batch_add_threshold_proof(b, [1, 2], 5)
    with pytest.raises(ValueError, match="data is not monotonic non-decreasing"):
        z.batch_add_consistency_proof(b, [3, 2])
    with pytest.raises(ValueError, match="new value must be greater than old value"):
        z.batch_add_improvement_proof(b, 5, 5)
    z.clear_batch(b)
    with pytest.raises(ValueError, match="Invalid batch ID"):
        z.get_batch_status(b)
    with pytest.raises(ValueError, match="Invalid batch ID"):
        z.process_batch(b)
    z.clear_batch(b)   # clearing an unknown batch is not an error (batch.rs:165-173)


def test_threshold_and_consistency_validation_messages():
    with pytest.raises(ValueError, match="^values cannot be empty$"):
        z.prove_threshold([], 5)
    with pytest.raises(ValueError, match="^integer overflow in sum calculation$"):
        z.prove_threshold([2**64 - 1, 1], 5)
    with pytest.raises(ValueError, match="^sum 30 is less than threshold 31$"):
        z.prove_threshold([10, 20], 31)
    with pytest.raises(ValueError, match="^data cannot be empty$"):
        z.prove_consistency([])
    with pytest.raises(ValueError, match="^data is not monotonic non-decreasing$"):
        z.prove_consistency([1, 3, 2])


def test_snark_validation_messages():
    with pytest.raises(ValueError, match="^values are not equal$"):
        z.prove_equality(1, 2)
    with pytest.raises(ValueError, match="^set cannot be empty$"):
        z.prove_membership(1, [])
    with pytest.raises(ValueError, match="^value 5 is not in the provided set$"):
        z.prove_membership(5, [1, 2, 3])
    with pytest.raises(ValueError, match="^set size 65 exceeds maximum allowed size 64$"):
        z.prove_membership(3, list(range(65)))
    with pytest.raises(TypeError, match="SNARK key directory cannot be empty"):
        z.set_snark_key_dir("")


def test_process_batch_consumes_the_batch_even_on_failure():
    b = z.create_proof_batch()
    z.batch_add_improvement_proof(b, 1, 2)
    with pytest.raises(RuntimeError):                            # no GPU in this tier: the product path fails loudly, no CPU fallback
        z.process_batch(b)
    with pytest.raises(ValueError, match="Invalid batch ID"):    # batch.rs:111-118
        z.process_batch(b)


def test_benchmark_rejects_unknown_type():
    with pytest.raises(ValueError, match="unsupported proof type: nope"):
        z.benchmark_proof_generation("nope", 1)


def test_shard_plan_is_contiguous_per_variant_and_balanced():
    """zkp_hip_plan_shards (pure host logic behind the multi-GPU batch calls): every variant's ops, in the caller's
    order, go to the shards in contiguous slices whose sizes differ by at most one; one shard = identity."""
    import ctypes
    import numpy as np
    from libzkp_amd import _native, workloads as wl
    L = _native.lib()
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    rng = np.random.default_rng(3)
    for n in (0, 1, 7, 103, 4096):
        ops, _, _ = wl.mixed_ops(max(n, 1), 5)
        ops = ops[:n].copy()
        if n > 20:                                            # uneven mix: a run of range ops and a few thresholds
            ops["kind"][:9] = wl.OP_RANGE
            ops["kind"][rng.choice(n, 5, replace=False)] = wl.OP_THRESHOLD
        for S in (1, 2, 3, 8):
            owner = np.full(max(n, 1), 99, dtype=np.uint32)
            assert L.zkp_hip_plan_shards(n, P(ops), S, P(owner)) == 0
            owner = owner[:n]
            assert (owner < S).all()
            for kind in range(1, 7):
                o = owner[ops["kind"] == kind]
                assert (np.diff(o.astype(np.int64)) >= 0).all()                 # contiguous slices in the caller's order
                sizes = np.bincount(o, minlength=S)
                assert sizes.max() - sizes.min() <= 1
            if S == 1:
                assert not owner.any()
    bad = np.zeros(1, dtype=wl.OP_DTYPE)
    bad["kind"] = 9
    assert L.zkp_hip_plan_shards(1, P(bad), 2, P(np.zeros(1, dtype=np.uint32))) == -3


def test_cache_and_metrics_bookkeeping():
    """performance.rs:24-215 semantics of the host-side cache / metrics behind clear_cache, get_cache_stats and
    get_performance_metrics (the proving itself needs a GPU and is covered in the -m gpu tier)."""
    from libzkp_amd import perf
    c = perf.ProofCache(max_size=2, ttl_seconds=3600)
    base = perf.METRICS.snapshot()
    assert c.get("a") is None                                 # miss
    c.put("a", b"1"); c.put("b", b"2")
    assert c.get("a") == b"1" and c.get("a") == b"1"          # a: access count 3, b: 1
    c.put("c", b"3")                                          # evicts the least frequently used entry (b)
    assert c.size() == 2 and c.get("b") is None and c.get("c") == b"3"
    t = perf.ProofCache(max_size=2, ttl_seconds=0)
    t.put("x", b"9")
    assert t.get("x") is None and t.size() == 0               # expired entries are dropped on access
    after = perf.METRICS.snapshot()
    assert after[2] - base[2] == 3 and after[3] - base[3] == 3
    k1, k2 = perf.generate_cache_key("range_proof", b"1:0:9"), perf.generate_cache_key("range_proof", b"1:0:8")
    assert k1 != k2 and k1.startswith("range_proof:") and len(k1) == len("range_proof:") + 64
    perf.METRICS.record_operation("range_proof", 0.0125)
    m = perf.performance_metrics()
    assert m["range_proof_count"] >= 1 and "avg_range_proof_time_ms" in m and m["total_operations"] >= 1 and 0.0 <= m["cache_hit_rate"] <= 1.0


def test_python_surface_has_every_reference_name():
    """python_api.rs:110-164 registers 49 functions; the Python mirror exposes each of them."""
    import libzkp_amd as z
    names = """prove_range verify_range prove_equality verify_equality verify_equality_with_commitment snark_commit_value prove_threshold
    verify_threshold prove_membership verify_membership prove_improvement verify_improvement prove_consistency verify_consistency
    create_composite_proof verify_composite_proof verify_composite_proof_integrity_only create_proof_with_metadata extract_proof_metadata
    clear_cache get_cache_stats get_performance_metrics benchmark_proof_generation_numeric prove_range_cached prove_equality_advanced
    verify_proofs_parallel benchmark_proof_generation prove_threshold_optimized validate_proof_chain get_proof_info set_snark_key_dir
    is_snark_setup_initialized create_proof_batch batch_add_range_proof batch_add_equality_proof batch_add_threshold_proof
    batch_add_membership_proof batch_add_improvement_proof batch_add_consistency_proof process_batch get_batch_status clear_batch
    set_batch_store_dir get_batch_store_dir list_batch_ids_in_store open_batch_from_store refresh_batch_from_store export_batch_to_file
    import_batch_from_file""".split()
    assert len(names) == 49 and all(callable(getattr(z, n)) for n in names)
