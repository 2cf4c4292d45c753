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
