"""The ops of ONE kind out of the metric's mixed batch, proved alone through zkp_hip_process_batch (for per-kernel times without
the other variants sharing the GPU): python tools/by_kind.py KIND [N_MIXED] [REPS]; KIND = 1 range, 2 equality, 4 membership,
5 improvement."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libzkp_amd import _native, workloads as wl
L = _native.lib()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
kind = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096; reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
_native.check(L.zkp_hip_init(0), "init")
for k, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
    blob = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    assert L.zkp_hip_groth16_load_key(k, blob, len(blob)) == 0, _native.last_error()
ops, lists, seeds = wl.mixed_ops(n, 5)
ix = np.nonzero(ops["kind"] == kind)[0]
sub = ops[ix].copy(); sd = np.ascontiguousarray(seeds.reshape(n, 32)[ix]).ravel()
if os.environ.get("SAME_OPS"):        # every op identical: all lanes of the MSM gather the same table entries (cache hits) -- a probe of how much of the kernel time is memory
    sub[:] = sub[0]; sd = np.tile(sd[:32], len(ix))
m = len(ix); cap = wl.max_output_bytes(sub)
out = np.zeros(cap, dtype=np.uint8); off = np.zeros(m + 1, dtype=np.uint64); st = np.zeros(m, dtype=np.int32)
ts = []
for _ in range(reps + 1):
    t0 = time.perf_counter(); rc = L.zkp_hip_process_batch(m, P(sub), P(lists), P(sd), P(out), cap, P(off), P(st)); ts.append(time.perf_counter() - t0)
    assert rc == 0 and not st.any(), _native.last_error()
print("kind %d: %d ops alone, min %.2f ms median %.2f ms (host buffers)" % (kind, m, min(ts[1:]) * 1e3, sorted(ts[1:])[len(ts[1:]) // 2] * 1e3))
L.zkp_hip_shutdown()
