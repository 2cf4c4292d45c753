"""A/B figures of one build (ZKP_HIP_LIB selects it): staged step times of the mixed batch at several sizes and of C2 / C3 on their own.
python tools/ab_steps.py [REPS]  -> one JSON line"""
import ctypes, json, os, statistics, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libzkp_amd import _native, workloads as wl
L = _native.lib()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 11
t0 = time.perf_counter()
_native.check(L.zkp_hip_init(0), "init")
t_init = time.perf_counter() - t0
for k, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
    blob = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    assert L.zkp_hip_groth16_load_key(k, blob, len(blob)) == 0, _native.last_error()


def staged(gen, n):
    ops, lists, seeds = gen(n) if gen is not wl.mixed_ops else gen(n, 5)
    h = ctypes.c_void_p()
    assert L.zkp_hip_batch_stage(n, P(ops), P(lists), P(seeds), ctypes.byref(h)) == 0, _native.last_error()
    for _ in range(3):
        assert L.zkp_hip_batch_prove(h) == 0, _native.last_error()
    ts = []
    for _ in range(reps):
        t = time.perf_counter(); assert L.zkp_hip_batch_prove(h) == 0; ts.append((time.perf_counter() - t) * 1e3)
    L.zkp_hip_batch_free(h)
    return round(statistics.median(ts), 3)


res = {"lib": os.path.basename(_native.LIB_PATH), "init_s": round(t_init, 2)}
for n in (512, 2048, 4096, 16384):
    res["mixed_%d_ms" % n] = staged(wl.mixed_ops, n)
res["C2_range_4096_ms"] = staged(wl.range_ops, 4096)
res["range_1024_ms"] = staged(wl.range_ops, 1024)
res["C3_equality_4096_ms"] = staged(wl.equality_ops, 4096)
print(json.dumps(res), flush=True)
L.zkp_hip_shutdown()
