"""How long the host takes to enqueue one mixed batch (zkp_hip_batch_prove_async returns when everything is queued) against the
whole step (enqueue + wait): python tools/enqueue_time.py [N] [REPS]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libzkp_amd import _native, workloads as wl
L = _native.lib()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
_native.check(L.zkp_hip_init(0), "init")
for k, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
    blob = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    assert L.zkp_hip_groth16_load_key(k, blob, len(blob)) == 0, _native.last_error()
ops, lists, seeds = wl.mixed_ops(n, 5)
h = ctypes.c_void_p()
assert L.zkp_hip_batch_stage(n, P(ops), P(lists), P(seeds), ctypes.byref(h)) == 0, _native.last_error()
for _ in range(3):
    assert L.zkp_hip_batch_prove(h) == 0
enq, tot = [], []
for _ in range(reps):
    t0 = time.perf_counter(); assert L.zkp_hip_batch_prove_async(h) == 0; t1 = time.perf_counter(); assert L.zkp_hip_batch_wait(h) == 0; t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print("enqueue ms: min %.2f median %.2f max %.2f | step ms: min %.2f median %.2f max %.2f" % (min(enq), sorted(enq)[len(enq) // 2], max(enq), min(tot), sorted(tot)[len(tot) // 2], max(tot)))
L.zkp_hip_batch_free(h)
L.zkp_hip_shutdown()
