"""Rates of the SURVEY section-8(f) "next" rows on one MI355X through the Python mirror (host buffers in and out, PCIe and the
Python marshalling included), best of 3 after a warm-up: proving at the four Bulletproofs bit widths (N4) and batched verification of all six schemes (N2).
Prints one JSON object; the committed copy is profiles/r01_next_rows.json."""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libzkp_amd as z
from libzkp_amd import _native, api
L = _native.lib()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
_native.check(L.zkp_hip_init(0), "init")
for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
    blob = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
    api.install_proving_key(kind, blob)
rng = np.random.default_rng(7)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def best(f, reps=3):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return min(ts), r


res = {"ops": n, "unit": "ms per batch (Python mirror, host buffers)", "prove_range_with_bits": {}, "prove_threshold_with_bits": {}, "verify": {}}
seeds = rng.integers(0, 256, 32 * n, dtype=np.uint8).tobytes()
proofs64 = None
for bits in (8, 16, 32, 64):
    cap = 2**bits - 1
    mn = np.zeros(n, dtype=np.uint64); mx = np.full(n, min(cap, 2**32), dtype=np.uint64)
    v = rng.integers(0, int(mx[0]), n, dtype=np.uint64, endpoint=True)
    dt, pr = best(lambda: z.prove_range_batch(v, mn, mx, seeds=seeds, n_bits=bits))
    res["prove_range_with_bits"][str(bits)] = {"ms": round(dt * 1e3, 2), "proofs_per_s": round(n / dt), "bytes": len(pr[0])}
    dtv, ok = best(lambda: z.verify_range_batch(pr, mn, mx))
    assert all(ok)
    res["verify"]["range_%d" % bits] = {"ms": round(dtv * 1e3, 2), "envelopes_per_s": round(n / dtv)}
    lists = [[int(x) for x in rng.integers(0, min(cap, 2**30) // 4 + 1, 3)] for _ in range(n)]
    thr = [max(0, sum(l) - int(rng.integers(0, min(cap, sum(l)), endpoint=True))) for l in lists]
    dt, tp = best(lambda: z.prove_threshold_batch(lists, thr, seeds=seeds, n_bits=bits))
    res["prove_threshold_with_bits"][str(bits)] = {"ms": round(dt * 1e3, 2), "proofs_per_s": round(n / dt), "bytes": len(tp[0])}
    if bits == 64:
        dtv, ok = best(lambda: z.verify_threshold_batch(tp, thr)); assert all(ok)
        res["verify"]["threshold"] = {"ms": round(dtv * 1e3, 2), "envelopes_per_s": round(n / dtv)}
data = [sorted(int(x) for x in rng.integers(0, 2**40, 4)) for _ in range(n)]
cp = z.prove_consistency_batch(data, seeds=seeds)
dtv, ok = best(lambda: z.verify_consistency_batch(cp)); assert all(ok)
res["verify"]["consistency_4_values"] = {"ms": round(dtv * 1e3, 2), "envelopes_per_s": round(n / dtv)}
olds = [int(x) for x in rng.integers(0, 2**40, n)]; news = [o + 1 + int(x) for o, x in zip(olds, rng.integers(0, 2**20, n))]
ip = z.prove_improvement_batch(olds, news)
dtv, ok = best(lambda: z.verify_improvement_batch(ip, olds)); assert all(ok)
res["verify"]["improvement"] = {"ms": round(dtv * 1e3, 2), "envelopes_per_s": round(n / dtv)}
vals = [int(x) for x in rng.integers(0, 2**62, n)]
ep = z.prove_equality_batch(vals, vals, seeds=seeds)
dtv, ok = best(lambda: api._verify_snark_envelopes(0, ep)); assert all(ok)
res["verify"]["equality"] = {"ms": round(dtv * 1e3, 2), "envelopes_per_s": round(n / dtv)}
m = min(n, 1024)
sets = [[int(x) for x in rng.choice(2**32, 16, replace=False)] for _ in range(m)]
mp = z.prove_membership_batch([s[3] for s in sets], sets, seeds=seeds[: 32 * m])
dtv, ok = best(lambda: z.verify_membership_batch(mp, sets)); assert all(ok)
res["verify"]["membership_16_of_%d" % m] = {"ms": round(dtv * 1e3, 2), "envelopes_per_s": round(m / dtv)}
print(json.dumps(res))
