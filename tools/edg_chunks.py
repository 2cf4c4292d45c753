"""Step time of range-only batches against the chunk count of the gather MSM launches (zkp_hip_set_window_budget(10000 + chunks) picks, for
every launch, the even layout closest to that many chunks): the data behind pick_layout's cost model.  python tools/edg_chunks.py"""
import ctypes, json, os, statistics, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libzkp_amd import _native, workloads as wl
L = _native.lib()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
_native.check(L.zkp_hip_init(0), "init")
for n in (256, 1024, 4096, 16384):
    ops, lists, seeds = wl.range_ops(n)
    h = ctypes.c_void_p()
    assert L.zkp_hip_batch_stage(n, P(ops), P(lists), P(seeds), ctypes.byref(h)) == 0, _native.last_error()
    row = {"range_ops": n, "rows": 2 * n}
    for want in (0, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024):
        L.zkp_hip_set_window_budget(10000 + want if want else 0)
        for _ in range(2):
            assert L.zkp_hip_batch_prove(h) == 0, _native.last_error()
        ts = []
        for _ in range(7):
            t = time.perf_counter(); assert L.zkp_hip_batch_prove(h) == 0; ts.append((time.perf_counter() - t) * 1e3)
        row["auto" if not want else str(want)] = round(statistics.median(ts), 3)
    L.zkp_hip_set_window_budget(0)
    L.zkp_hip_batch_free(h)
    print(json.dumps(row), flush=True)
L.zkp_hip_shutdown()
