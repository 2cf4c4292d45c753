"""BASELINE config 5 shape on one GPU: a mixed batch (range / equality / membership / improvement interleaved, i mod 4)
through the Python mirror of process_batch: order preservation, per-variant byte parity with the oracles on a sample,
and verifier acceptance."""
import ctypes
import hashlib
import os

import numpy as np
import pytest

from oracle.py import groth16 as g
from oracle.py import stark
from util import oracle_prove

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SS = bytes(range(32))


def test_mixed_batch_c5_shape(oracle_c, tmp_path):
    import libzkp_amd as z
    from libzkp_amd import _native
    L = _native.lib()
    _native.check(L.zkp_hip_init(0), "zkp_hip_init")
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):   # the committed test keys (setup seed 0..31)
        blob = open(os.path.join(GOLD, name), "rb").read()
        assert L.zkp_hip_groth16_load_key(kind, blob, len(blob)) == 0, _native.last_error()
    import libzkp_amd.api as api
    with api._snark_lock:                      # the golden keys are now the loaded ones, whatever earlier tests did
        api._keys_loaded[0] = True
        api._keys_loaded[1] = True
    rng = np.random.default_rng(5)
    n = 256
    b = z.create_proof_batch()
    ops = []
    for i in range(n):
        k = i % 4
        if k == 0:
            v = int(rng.integers(0, 2**32, endpoint=True)); z.batch_add_range_proof(b, v, 0, 2**32); ops.append(("range", v))
        elif k == 1:
            a = int(rng.integers(0, 2**63)); z.batch_add_equality_proof(b, a, a); ops.append(("equality", a))
        elif k == 2:
            s = [int(x) for x in rng.choice(2**32, 16, replace=False)]; v = s[i % 16]
            z.batch_add_membership_proof(b, v, s); ops.append(("membership", v, s))
        else:
            old = int(rng.integers(0, 2**63)); new = old + 1 + int(rng.integers(0, 2**32))
            z.batch_add_improvement_proof(b, old, new); ops.append(("improvement", old, new))
    st = z.get_batch_status(b)
    assert st["total_operations"] == n and st["range_proofs"] == st["equality_proofs"] == st["membership_proofs"] == st["improvement_proofs"] == n // 4
    seeds = b"".join(hashlib.sha256((5).to_bytes(8, "little") + i.to_bytes(8, "little")).digest() for i in range(n))
    proofs = z.process_batch(b, seeds=seeds)
    assert len(proofs) == n
    scheme = {"range": 1, "equality": 2, "membership": 4, "improvement": 5}
    for op, p in zip(ops, proofs):
        assert p[0] == 2 and p[1] == scheme[op[0]]                     # order preserved: envelope scheme ids line up with the ops
    # improvement: bit-exact (deterministic) for all
    for i in range(3, n, 4):
        assert proofs[i] == stark.prove_improvement(ops[i][1], ops[i][2])
    # range: bit-exact against the C oracle under the same seeds
    ridx = list(range(0, n, 4))
    vals = np.array([ops[i][1] for i in ridx], dtype=np.uint64)
    sd = np.frombuffer(b"".join(seeds[32 * i:32 * i + 32] for i in ridx), dtype=np.uint8).copy()
    rc, ref, lens, stt = oracle_prove(oracle_c, vals, np.zeros(len(ridx), dtype=np.uint64), np.full(len(ridx), 2**32, dtype=np.uint64), sd, threads=8)
    assert rc == 0 and all(proofs[i] == ref[k].tobytes() for k, i in enumerate(ridx))
    # Groth16: pairing verification of a sample (the prover's r, s come from the seeds)
    for i in (1, 5, n - 3):
        assert g.verify_equality_with_commitment(proofs[i], g.commit_value_snark(ops[i][1]), SS)
    i = 2
    assert g.verify_membership(proofs[i], ops[i][2], SS)


def test_c_abi_process_batch_equals_per_variant_calls(oracle_c):
    """zkp_hip_process_batch (the compiled process_batch) against the per-variant entry points on the same seeds,
    including threshold / consistency ops, a failing op and the too-small-buffer protocol."""
    import ctypes
    import libzkp_amd.api as api
    from libzkp_amd import _native
    from util import P
    L = _native.lib()
    _native.check(L.zkp_hip_init(0), "zkp_hip_init")
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        blob = open(os.path.join(GOLD, name), "rb").read()
        assert L.zkp_hip_groth16_load_key(kind, blob, len(blob)) == 0
    with api._snark_lock:
        api._keys_loaded[0] = api._keys_loaded[1] = True
    ops = [("range", 7, 0, 100), ("threshold", (10, 20, 30), 50), ("improvement", 3, 9), ("consistency", (1, 5, 5, 9)),
           ("equality", 42, 42), ("membership", 25, (10, 20, 25, 30)), ("range", 2**32, 0, 2**32), ("threshold", (2**40,), 1)]
    seeds = bytes((11 * k + 5) % 256 for k in range(32 * len(ops)))
    got = api.process_ops(ops, seeds)                              # one C call
    want = api.prove_ops(ops, seeds)                               # one call per variant
    assert got == want and all(len(p) > 0 for p in got)
    assert [p[1] for p in got] == [1, 3, 5, 6, 2, 4, 1, 3]         # scheme ids in the caller's order
    assert api.verify_range(got[0], 0, 100) and api.verify_range(got[6], 0, 2**32)
    # a failing op (value outside its range) fails the batch and is reported per item
    n = 3
    arr = (_native.Op * n)()
    arr[0].kind, arr[0].a, arr[0].b, arr[0].c = 1, 5, 0, 10
    arr[1].kind, arr[1].a, arr[1].b, arr[1].c = 1, 50, 0, 10
    arr[2].kind, arr[2].a, arr[2].b = 5, 1, 2
    out = np.zeros(8192, dtype=np.uint8); off = np.zeros(n + 1, dtype=np.uint64); st = np.zeros(n, dtype=np.int32)
    rc = L.zkp_hip_process_batch(n, ctypes.byref(arr), None, seeds[:96], P(out), 8192, P(off), P(st))
    assert rc == 1 and list(st) == [0, 1, 0] and off[1] == off[2] == 1478
    # too small a buffer: -3 and the required size
    arr[1].a = 5
    rc = L.zkp_hip_process_batch(n, ctypes.byref(arr), None, seeds[:96], P(out), 100, P(off), P(st))
    assert rc == -3 and int(off[n]) == 2 * 1478 + len(stark.prove_improvement(1, 2))
    rc = L.zkp_hip_process_batch(n, ctypes.byref(arr), None, seeds[:96], P(out), 8192, P(off), P(st))
    assert rc == 0 and out[int(off[2]):int(off[3])].tobytes() == stark.prove_improvement(1, 2)
