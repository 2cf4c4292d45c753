"""Launch trace of an equality-only batch (BASELINE configs[2] as a 4096-op batch) on one MI355X (development aid): stages the batch,
proves it a few times with ZKP_HIP_TRACE set and prints the last batch's timeline per stream.  Usage: ZKP_HIP_TRACE=/tmp/t.json python3 tools/c3_trace.py [n]"""
import ctypes, os, sys, time, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libzkp_amd import _native, workloads as wl
L = _native.lib()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
_native.check(L.zkp_hip_init(0), "init")
for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
    blob = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    _native.check(L.zkp_hip_groth16_load_key(kind, blob, len(blob)), "load_key")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ops, lists, seeds = wl.equality_ops(n)
h = ctypes.c_void_p()
_native.check(L.zkp_hip_batch_stage(n, P(ops), P(lists), P(seeds), ctypes.byref(h)), "stage")
ts = []
for _ in range(6):
    t = time.perf_counter(); _native.check(L.zkp_hip_batch_prove(h), "prove"); ts.append(time.perf_counter() - t)
print("equality-only batch of %d: %.2f ms (best of 6)" % (n, min(ts) * 1e3))
L.zkp_hip_batch_free(h)
f = os.environ.get("ZKP_HIP_TRACE")
if f: subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_timeline.py"), f])
