"""Host-buffer (PCIe-inclusive) throughput of the C ABI: prove_range batch, the C5-shaped mixed process_batch, and
verify_range (development aid; bench.py reports the HBM-resident headline number)."""
import ctypes, hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libzkp_amd import _native
import libzkp_amd.api as api
L = _native.lib()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
_native.check(L.zkp_hip_init(0), "init")
for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
    blob = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    assert L.zkp_hip_groth16_load_key(kind, blob, len(blob)) == 0
with api._snark_lock:
    api._keys_loaded[0] = api._keys_loaded[1] = True
rng = np.random.default_rng(1)
n = 4096
v = rng.integers(0, 2**32, n, dtype=np.uint64, endpoint=True); mn = np.zeros(n, dtype=np.uint64); mx = np.full(n, 2**32, dtype=np.uint64)
seeds = rng.integers(0, 256, 32 * n, dtype=np.uint8)
out = np.zeros((n, 1478), dtype=np.uint8); ln = np.zeros(n, dtype=np.uint32); st = np.zeros(n, dtype=np.int32)
for it in range(4):
    t0 = time.perf_counter(); rc = L.zkp_hip_prove_range_batch(n, P(v), P(mn), P(mx), 64, P(seeds), P(out), 1478, P(ln), P(st)); dt = time.perf_counter() - t0
    print("prove_range host buffers n=%d: %.2f ms -> %.0f ops/s rc=%d" % (n, dt * 1e3, n / dt, rc))
ok = np.zeros(n, dtype=np.uint8)
for it in range(3):
    t0 = time.perf_counter(); rc = L.zkp_hip_verify_range_batch(n, P(out), 1478, P(ln), P(mn), P(mx), P(ok)); dt = time.perf_counter() - t0
    print("verify_range host buffers n=%d: %.2f ms -> %.0f envelopes/s all ok=%s" % (n, dt * 1e3, n / dt, bool((ok == 1).all())))
# C5: 16384 mixed ops
N = 16384
ops = []
for i in range(N):
    k = i % 4
    if k == 0:
        ops.append(("range", int(rng.integers(0, 2**32, endpoint=True)), 0, 2**32))
    elif k == 1:
        a = int(rng.integers(0, 2**63)); ops.append(("equality", a, a))
    elif k == 2:
        s = [int(x) for x in rng.choice(2**32, 16, replace=False)]; ops.append(("membership", s[i % 16], tuple(s)))
    else:
        o = int(rng.integers(0, 2**63)); ops.append(("improvement", o, o + 1 + int(rng.integers(0, 2**32))))
sd = b"".join(hashlib.sha256((5).to_bytes(8, "little") + i.to_bytes(8, "little")).digest() for i in range(N))
for it in range(3):
    t0 = time.perf_counter(); proofs = api.process_ops(ops, sd); dt = time.perf_counter() - t0
    print("process_batch C5 (16384 mixed ops, Python marshalling + one C call): %.1f ms -> %.0f proofs/s" % (dt * 1e3, N / dt))
