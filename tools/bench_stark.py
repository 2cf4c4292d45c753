import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libzkp_amd import _native
L = _native.lib()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(3)
olds = rng.integers(0, 2**63, n, dtype=np.uint64)
news = olds + 1 + rng.integers(0, 2**32, n, dtype=np.uint64)
stride = int(L.zkp_hip_improvement_max_bytes())
out = np.zeros((n, stride), dtype=np.uint8); ln = np.zeros(n, dtype=np.uint32); st = np.zeros(n, dtype=np.int32)
for it in range(4):
    t0 = time.time(); rc = L.zkp_hip_prove_improvement_batch(n, P(olds), P(news), P(out), stride, P(ln), P(st)); dt = time.time() - t0
    print("improvement n", n, "rc", rc, "%.2f ms -> %.0f proofs/s" % (dt * 1e3, n / dt), "ok" if (st == 0).all() else "FAIL", "mean len %.0f" % ln.mean())
