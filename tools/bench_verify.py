"""Range-envelope verification rate through zkp_hip_verify_range_batch (host buffers in, verdicts out): envelopes proved on the
GPU, then verified with the batch check (one random-linear-combination MSM over the whole batch, bpv_impl.inc) and with the
per-job check, at several batch sizes; also a batch with one tampered envelope, which the batch check must hand to the per-job
path.  Prints one JSON object (profiles/r02_verify_rates.json is this output)."""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libzkp_amd import _native, workloads as wl
L = _native.lib()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
sizes = [int(a) for a in sys.argv[1:]] or [1024, 2048, 4096, 16384]


def run(n, buf, ln, lo, hi, ok):
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); rc = L.zkp_hip_verify_range_batch(n, P(buf), 1478, P(ln), P(lo), P(hi), P(ok)); ts.append(time.perf_counter() - t0)
        assert rc == 0
    return sorted(ts)[len(ts) // 2] * 1e3


rows = []
for n in sizes:
    ops, _, seeds = wl.range_ops(n, 1)
    v, lo, hi = ops["a"].copy(), ops["b"].copy(), ops["c"].copy()
    out = np.zeros((n, 1478), dtype=np.uint8); ln = np.zeros(n, dtype=np.uint32); st = np.zeros(n, dtype=np.int32)
    assert L.zkp_hip_prove_range_batch(n, P(v), P(lo), P(hi), 64, P(seeds), P(out), 1478, P(ln), P(st)) == 0
    ok = np.zeros(n, dtype=np.uint8)
    bad = out.copy(); bad[n // 2, 700] ^= 4
    row = {"envelopes": n}
    for mode, env in (("batch_check", {"ZKP_HIP_BATCH_VERIFY_MIN": "1"}), ("per_job", {"ZKP_HIP_NO_BATCH_VERIFY": "1"})):
        os.environ.update(env)
        t = run(n, out, ln, lo, hi, ok); assert ok.all()
        tb = run(n, bad, ln, lo, hi, ok); assert ok.sum() == n - 1 and ok[n // 2] == 0
        for k in env:
            del os.environ[k]
        row[mode] = {"ms": round(t, 3), "envelopes_per_s": round(n / t * 1e3), "ms_with_one_tampered": round(tb, 3)}
    rows.append(row)
print(json.dumps({"tool": "tools/bench_verify.py", "entry": "zkp_hip_verify_range_batch", "timing": "host wall clock, median of 7, host buffers", "rows": rows}))
L.zkp_hip_shutdown()
