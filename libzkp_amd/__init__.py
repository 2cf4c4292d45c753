"""libzkp_amd -- MI355X-native proving backend for libzkp's proving hot path.

Python surface mirroring the names of the reference's PyO3 module (/root/reference/src/python_api.rs:110-164)
for the path in scope: the six prove_* operations and the batch driver.  Everything is computed by the
HIP library behind the C ABI of include/libzkp_hip.h; there is no CPU fallback.
"""
from .api import (  # noqa: F401
    prove_range, prove_range_with_bits, prove_threshold_with_bits, prove_range_batch, verify_range, verify_range_batch, verify_threshold, verify_threshold_batch, verify_improvement, verify_improvement_batch, verify_consistency, verify_consistency_batch, verify_equality, verify_equality_with_commitment,
    verify_equality_with_commitment_batch, verify_membership, verify_membership_batch, prove_threshold, prove_threshold_batch, prove_consistency, prove_consistency_batch,
    prove_equality, prove_equality_batch, prove_equality_advanced, prove_membership, prove_membership_batch, prove_improvement,
    prove_improvement_batch, snark_commit_value,
    snark_commit_value_batch, set_snark_key_dir, is_snark_setup_initialized,
    create_proof_batch, batch_add_range_proof, batch_add_equality_proof,
    batch_add_threshold_proof, batch_add_membership_proof, batch_add_improvement_proof, batch_add_consistency_proof,
    process_batch, get_batch_status, clear_batch, benchmark_proof_generation, benchmark_proof_generation_numeric,
    open_batch_from_store, refresh_batch_from_store, export_batch_to_file, import_batch_from_file,
    ZkpBackendError, shutdown, clear_cache, get_cache_stats, get_performance_metrics, prove_range_cached, prove_threshold_optimized,
)
from .batch_store import set_batch_store_dir, get_batch_store_dir, list_batch_ids_in_store  # noqa: F401
from .composite import (  # noqa: F401
    create_composite_proof, verify_composite_proof, verify_composite_proof_integrity_only, create_proof_with_metadata,
    extract_proof_metadata, verify_proofs_parallel, validate_proof_chain, get_proof_info,
)
from ._native import NativeError  # noqa: F401

__all__ = [
    "prove_range", "prove_range_with_bits", "prove_threshold_with_bits", "prove_range_batch", "verify_range", "verify_range_batch", "verify_threshold", "verify_threshold_batch", "verify_improvement", "verify_improvement_batch", "verify_consistency", "verify_consistency_batch", "verify_equality", "verify_equality_with_commitment",
    "verify_equality_with_commitment_batch", "verify_membership", "verify_membership_batch", "prove_threshold", "prove_threshold_batch", "prove_consistency", "prove_consistency_batch",
    "prove_equality", "prove_equality_batch", "prove_equality_advanced", "prove_membership", "prove_membership_batch", "prove_improvement",
    "prove_improvement_batch", "snark_commit_value",
    "snark_commit_value_batch", "set_snark_key_dir", "is_snark_setup_initialized",
    "create_proof_batch", "batch_add_range_proof", "batch_add_equality_proof",
    "batch_add_threshold_proof", "batch_add_membership_proof", "batch_add_improvement_proof", "batch_add_consistency_proof",
    "process_batch", "get_batch_status", "clear_batch", "benchmark_proof_generation", "benchmark_proof_generation_numeric",
    "create_composite_proof", "verify_composite_proof", "verify_composite_proof_integrity_only", "create_proof_with_metadata",
    "extract_proof_metadata", "verify_proofs_parallel", "validate_proof_chain", "get_proof_info",
    "open_batch_from_store", "refresh_batch_from_store", "export_batch_to_file", "import_batch_from_file",
    "set_batch_store_dir", "get_batch_store_dir", "list_batch_ids_in_store",
    "NativeError", "ZkpBackendError", "shutdown",
    "clear_cache", "get_cache_stats", "get_performance_metrics", "prove_range_cached", "prove_threshold_optimized",
]
