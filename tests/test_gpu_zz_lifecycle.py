"""zkp_hip_shutdown / re-initialisation: a full init-work-shutdown cycle returns its device memory, and a re-initialised
library reproduces the same bytes and still verifies what it proved before (the Groth16 key is reinstalled, not regenerated).
Runs last (file name) so that the other modules' module-scoped fixtures are not torn down under them."""
import ctypes

import numpy as np
import pytest

from util import P, oracle_prove, outputs, workload

pytestmark = pytest.mark.gpu


def _free_bytes(L):
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    f = L.hipMemGetInfo
    f.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    f.restype = ctypes.c_int
    assert f(ctypes.byref(free), ctypes.byref(total)) == 0
    return free.value


def _work(z, L, seeds_eq):
    n = 700
    v, mn, mx, seeds = workload(n, 21)
    out, lens, st = outputs(n)
    assert L.zkp_hip_prove_range_batch(n, P(v), P(mn), P(mx), 64, P(seeds), P(out), 1478, P(lens), P(st)) == 0
    ok = np.zeros(n, dtype=np.uint8)
    assert L.zkp_hip_verify_range_batch(n, P(out), 1478, P(lens), P(mn), P(mx), P(ok)) == 0 and ok.all()
    p8 = z.prove_range_batch([3, 200], [0, 100], [9, 300], seeds=bytes(64), n_bits=8)
    eq = z.prove_equality_batch([5, 77], [5, 77], seeds=seeds_eq)
    imp = z.prove_improvement_batch([1, 10], [2, 1000])
    return out.copy(), p8, eq, imp


def test_cycle_releases_memory_and_reproduces(oracle_c):
    import libzkp_amd as z
    from libzkp_amd import _native
    L = _native.lib()
    _native.check(L.zkp_hip_init(0), "zkp_hip_init")
    seeds_eq = bytes(range(64))
    first = _work(z, L, seeds_eq)
    z.shutdown()
    second = _work(z, L, seeds_eq)                                    # initialises again
    assert (first[0] == second[0]).all() and first[1] == second[1] and first[3] == second[3]
    assert first[2] == second[2]                                      # same Groth16 key after the reinstall, same seeds
    assert all(z.verify_equality(p, x, x) for p, x in zip(first[2], (5, 77)))     # proofs from before the shutdown
    rc, ref, _, _ = oracle_prove(oracle_c, *workload(700, 21))
    assert rc == 0 and (ref == second[0]).all()
    z.shutdown()
    # The HIP runtime keeps the scratch (private-segment) reservation of the queues the first cycles created -- about 5 GB for
    # the pairing kernels' 12 KB per lane -- so the steady state is compared: a third cycle must leave nothing more behind.
    free_b = _free_bytes(L)
    _work(z, L, seeds_eq)
    z.shutdown()
    free_c = _free_bytes(L)
    assert free_c + (64 << 20) >= free_b, (free_b, free_c)
    L.zkp_hip_shutdown()                                              # idempotent
    _native.check(L.zkp_hip_init(0), "zkp_hip_init")                  # leave the library usable
