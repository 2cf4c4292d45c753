"""Additions per second of the three MSM kernels with each variant ALONE on the GPU (C2 = 4096 range ops for the ed25519 kernel, C3 = 4096
equality ops for the two BN254 kernels), from the library's own event pairs around every launch (zkp_hip_profile_read_kernel).
ZKP_HIP_LIB selects the build.  python tools/msm_rates.py -> one JSON line"""
import ctypes, json, os, sys
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
res = {"lib": os.path.basename(_native.LIB_PATH)}
for name, gen, n in (("C2_range_4096", wl.range_ops, 4096), ("C3_equality_4096", wl.equality_ops, 4096), ("membership_1024", None, 1024)):
    if gen is None:
        ops, lists, seeds = wl.mixed_ops(4 * n, 5)
        keep = ops["kind"] == 4
        ops, seeds = ops[keep].copy(), np.ascontiguousarray(seeds.reshape(-1, 32)[keep]).ravel()
    else:
        ops, lists, seeds = gen(n)
    h = ctypes.c_void_p()
    assert L.zkp_hip_batch_stage(len(ops), P(ops), P(lists), P(seeds), ctypes.byref(h)) == 0, _native.last_error()
    for _ in range(3):
        L.zkp_hip_batch_prove(h)
    L.zkp_hip_profile_enable(1)
    for k in range(3):
        L.zkp_hip_profile_read_kernel(k, None, None, None, 1)
    reps = 8
    for _ in range(reps):
        L.zkp_hip_batch_prove(h)
    out = {}
    for k, kn in enumerate(("ed25519", "bn254_g1", "bn254_g2")):
        ms, launches, adds = ctypes.c_double(), ctypes.c_uint64(), ctypes.c_uint64()
        L.zkp_hip_profile_read_kernel(k, ctypes.byref(ms), ctypes.byref(launches), ctypes.byref(adds), 1)
        if launches.value:
            out[kn] = {"ms_per_batch": round(ms.value / reps, 4), "launches_per_batch": launches.value // reps, "g_adds_per_s": round(adds.value / (ms.value * 1e-3) / 1e9, 2)}
    L.zkp_hip_profile_enable(0)
    L.zkp_hip_batch_free(h)
    res[name] = out
print(json.dumps(res), flush=True)
L.zkp_hip_shutdown()
