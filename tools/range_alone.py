"""Range-only batches through the host-buffer entry and through the staged scheduler, every repetition printed (bimodality probe):
python tools/range_alone.py [N] [REPS] [mixed_first]"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libzkp_amd import _native, workloads as wl
L = _native.lib()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
mixed_first = len(sys.argv) > 3 and sys.argv[3] == "1"
staged_first = len(sys.argv) > 4 and sys.argv[4] == "1"
_native.check(L.zkp_hip_init(0), "init")
if mixed_first:
    for k, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        blob = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
        assert L.zkp_hip_groth16_load_key(k, blob, len(blob)) == 0, _native.last_error()
    ops, lists, seeds = wl.mixed_ops(4096, 5)
    h = ctypes.c_void_p()
    assert L.zkp_hip_batch_stage(4096, P(ops), P(lists), P(seeds), ctypes.byref(h)) == 0
    for _ in range(5):
        assert L.zkp_hip_batch_prove(h) == 0
    L.zkp_hip_batch_free(h)
ops, lists, seeds = wl.range_ops(n)
v, mn, mx = ops["a"].copy(), ops["b"].copy(), ops["c"].copy()
out, ln, st = np.zeros((n, 1478), dtype=np.uint8), np.zeros(n, dtype=np.uint32), np.zeros(n, dtype=np.int32)
def host_entry():
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); assert L.zkp_hip_prove_range_batch(n, P(v), P(mn), P(mx), 64, P(seeds), P(out), 1478, P(ln), P(st)) == 0; ts.append((time.perf_counter() - t0) * 1e3)
    print("host-buffer entry, %d ops:" % n, " ".join("%.2f" % t for t in ts))
h = ctypes.c_void_p()
assert L.zkp_hip_batch_stage(n, P(ops), P(lists), P(seeds), ctypes.byref(h)) == 0
def staged():
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); assert L.zkp_hip_batch_prove(h) == 0; ts.append((time.perf_counter() - t0) * 1e3)
    print("staged scheduler,  %d ops:" % n, " ".join("%.2f" % t for t in ts))
for f in ((staged, host_entry, staged) if staged_first else (host_entry, staged, host_entry)):
    f()
L.zkp_hip_batch_free(h)
L.zkp_hip_shutdown()
