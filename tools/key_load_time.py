"""Time zkp_hip_groth16_load_key for both circuits (window-table construction on the GPU): python tools/key_load_time.py [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libzkp_amd import _native
L = _native.lib()
_native.check(L.zkp_hip_init(0), "init")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
for r in range(reps):
    ts = []
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        blob = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
        t0 = time.perf_counter(); assert L.zkp_hip_groth16_load_key(kind, blob, len(blob)) == 0, _native.last_error(); ts.append(time.perf_counter() - t0)
    print("load_key: equality %.3f s, membership %.3f s, both %.3f s" % (ts[0], ts[1], sum(ts)))
L.zkp_hip_shutdown()
