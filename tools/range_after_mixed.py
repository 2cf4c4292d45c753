"""The trap of round 2: a range-only batch through the HOST-BUFFER entry, timed after a mixed batch has created every stream and
hardware queue the library uses.  Prints one JSON line: ms per call of zkp_hip_prove_range_batch at 1024 and 4096 ops (median of 11), the staged
mixed batch beside it.  Run once per priority setting (the switches are read once per process)."""
import ctypes, json, os, statistics, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libzkp_amd import _native, workloads as wl
L = _native.lib()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
_native.check(L.zkp_hip_init(0), "init")
for k, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
    blob = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    assert L.zkp_hip_groth16_load_key(k, blob, len(blob)) == 0, _native.last_error()
ops, lists, seeds = wl.mixed_ops(4096, 5)
h = ctypes.c_void_p()
assert L.zkp_hip_batch_stage(4096, P(ops), P(lists), P(seeds), ctypes.byref(h)) == 0
ts = []
for i in range(14):
    t = time.perf_counter(); assert L.zkp_hip_batch_prove(h) == 0; ts.append((time.perf_counter() - t) * 1e3)
res = {"priorities": {k: os.environ.get(k) for k in ("ZKP_HIP_BP_PRIORITY", "ZKP_HIP_G16_SIDE2_PRIORITY")}, "mixed_4096_staged_ms": round(statistics.median(ts[3:]), 3)}
for n in (1024, 4096):
    o, _, sd = wl.range_ops(n, 1)
    v, lo, hi = o["a"].copy(), o["b"].copy(), o["c"].copy()
    out = np.zeros((n, 1478), dtype=np.uint8); ln = np.zeros(n, dtype=np.uint32); st = np.zeros(n, dtype=np.int32)
    ts = []
    for i in range(14):
        t = time.perf_counter(); assert L.zkp_hip_prove_range_batch(n, P(v), P(lo), P(hi), 64, P(sd), P(out), 1478, P(ln), P(st)) == 0; ts.append((time.perf_counter() - t) * 1e3)
    res["prove_range_batch_host_%d_ms" % n] = round(statistics.median(ts[3:]), 3)
    # interleave: a mixed batch between the range calls, as round 2's bench did
    ts = []
    for i in range(8):
        assert L.zkp_hip_batch_prove(h) == 0
        t = time.perf_counter(); assert L.zkp_hip_prove_range_batch(n, P(v), P(lo), P(hi), 64, P(sd), P(out), 1478, P(ln), P(st)) == 0; ts.append((time.perf_counter() - t) * 1e3)
    res["prove_range_batch_host_%d_after_each_mixed_ms" % n] = round(statistics.median(ts[2:]), 3)
L.zkp_hip_batch_free(h)
print(json.dumps(res), flush=True)
L.zkp_hip_shutdown()
