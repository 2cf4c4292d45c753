"""Mixed batch (C5's i mod 4 mix) through zkp_hip_process_batch with host buffers, next to the four variants proved
separately through their own entry points on the same ops -- the figure the scheduler has to beat (sum of the parts)."""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libzkp_amd import _native, workloads as wl  # noqa: E402

L = _native.lib()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if os.environ.get("SHARDS_ON_ONE_GPU"):          # the same GPU registered k times: k independent contexts, each proving a 1/k slice of every variant
    _native.init_devices([0] * int(os.environ["SHARDS_ON_ONE_GPU"]))
else:
    _native.check(L.zkp_hip_init(0), "init")
for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
    blob = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    assert L.zkp_hip_groth16_load_key(kind, blob, len(blob)) == 0, _native.last_error()
ops, lists, seeds = wl.mixed_ops(n, 5)
cap = wl.max_output_bytes(ops)
out = np.zeros(cap, dtype=np.uint8)
off = np.zeros(n + 1, dtype=np.uint64)
st = np.zeros(n, dtype=np.int32)


def mixed():
    rc = L.zkp_hip_process_batch(n, P(ops), P(lists), P(seeds), P(out), cap, P(off), P(st))
    assert rc == 0, (rc, _native.last_error())


def timeit(f, reps=reps):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3


best, med = timeit(mixed)
print("mixed %d ops, host buffers in and out: best %.2f ms, median %.2f ms -> %.0f proofs/s; %d output bytes" % (n, best, med, n / (med * 1e-3), int(off[n])))
h = ctypes.c_void_p()
assert L.zkp_hip_batch_stage(n, P(ops), P(lists), P(seeds), ctypes.byref(h)) == 0, _native.last_error()
best, med = timeit(lambda: L.zkp_hip_batch_prove(h))
print("mixed %d ops, staged in HBM, proofs left in HBM: best %.2f ms, median %.2f ms -> %.0f proofs/s" % (n, best, med, n / (med * 1e-3)))
L.zkp_hip_batch_free(h)

# the four variants on their own
sd = seeds.reshape(n, 32)
parts = {}
k = ops["kind"]
r = ops[k == wl.OP_RANGE]; rs = np.ascontiguousarray(sd[k == wl.OP_RANGE]); m = len(r)
ro, rl, rst = np.zeros((m, 1478), dtype=np.uint8), np.zeros(m, dtype=np.uint32), np.zeros(m, dtype=np.int32)
va, vb, vc = r["a"].copy(), r["b"].copy(), r["c"].copy()
parts["range"] = timeit(lambda: L.zkp_hip_prove_range_batch(m, P(va), P(vb), P(vc), 64, P(rs), P(ro), 1478, P(rl), P(rst)))
e = ops[k == wl.OP_EQUALITY]; es = np.ascontiguousarray(sd[k == wl.OP_EQUALITY]); m2 = len(e)
eo, el, est = np.zeros((m2, 298), dtype=np.uint8), np.zeros(m2, dtype=np.uint32), np.zeros(m2, dtype=np.int32)
ea = e["a"].copy()
parts["equality"] = timeit(lambda: L.zkp_hip_prove_equality_batch(m2, P(ea), P(ea), P(es), P(eo), 298, P(el), P(est)))
mm = ops[k == wl.OP_MEMBERSHIP]; ms = np.ascontiguousarray(sd[k == wl.OP_MEMBERSHIP]); m3 = len(mm)
stride = wl.membership_bytes(16)
mo, ml, mst = np.zeros((m3, stride), dtype=np.uint8), np.zeros(m3, dtype=np.uint32), np.zeros(m3, dtype=np.int32)
ma, mc = mm["a"].copy(), mm["count"].copy()
parts["membership"] = timeit(lambda: L.zkp_hip_prove_membership_batch(m3, P(ma), P(lists), P(mc), P(ms), P(mo), stride, P(ml), P(mst)))
im = ops[k == wl.OP_IMPROVEMENT]; m4 = len(im)
io_, il, ist = np.zeros((m4, 3527), dtype=np.uint8), np.zeros(m4, dtype=np.uint32), np.zeros(m4, dtype=np.int32)
ia, ib = im["a"].copy(), im["b"].copy()
parts["improvement"] = timeit(lambda: L.zkp_hip_prove_improvement_batch(m4, P(ia), P(ib), P(io_), 3527, P(il), P(ist)))
for name, (b, md) in parts.items():
    print("  %-12s %5d ops alone: best %.2f ms, median %.2f ms" % (name, n // 4, b, md))
print("sum of the parts (median): %.2f ms" % sum(v[1] for v in parts.values()))
L.zkp_hip_shutdown()
