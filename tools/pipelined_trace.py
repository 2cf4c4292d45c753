"""Launch traces of two mixed batches kept in flight (zkp_hip_batch_prove_async on two staged copies alternately, as bench.py's
two_batches_in_flight leg does) on one MI355X (development aid).  Usage: ZKP_HIP_TRACE=/tmp/t.json python3 tools/pipelined_trace.py"""
import ctypes, os, sys, time, json
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
n = 4096
ops, lists, seeds = wl.mixed_ops(n)
hs = []
for _ in range(2):
    h = ctypes.c_void_p(); _native.check(L.zkp_hip_batch_stage(n, P(ops), P(lists), P(seeds), ctypes.byref(h)), "stage"); hs.append(h)
def single(k):
    t = time.perf_counter()
    for _ in range(k): _native.check(L.zkp_hip_batch_prove(hs[0]), "prove")
    return (time.perf_counter() - t) / k * 1e3
def pipelined(k):
    t = time.perf_counter()
    _native.check(L.zkp_hip_batch_prove_async(hs[0]), "async")
    for i in range(1, k):
        _native.check(L.zkp_hip_batch_prove_async(hs[i & 1]), "async")
        _native.check(L.zkp_hip_batch_wait(hs[(i - 1) & 1]), "wait")
    _native.check(L.zkp_hip_batch_wait(hs[(k - 1) & 1]), "wait")
    return (time.perf_counter() - t) / k * 1e3
single(3); pipelined(4)
print("single %.2f ms/step, pipelined %.2f ms/step" % (single(12), pipelined(12)))
f = os.environ.get("ZKP_HIP_TRACE")
if f:
    lines = [json.loads(x) for x in open(f) if x.strip()]
    for k in (len(lines) - 3, len(lines) - 2):
        recs = lines[k]
        print("== batch %d: %d launches, done at %.3f ms" % (k, len(recs), max(r[3] for r in recs)))
        for s in sorted(set(r[1] for r in recs)):
            rs = [r for r in recs if r[1] == s]
            print("  stream %d: first reached %.3f, last done %.3f; big: %s" % (s, min(r[2] for r in rs), max(r[3] for r in rs),
                  ", ".join("%s %.2f-%.2f" % (r[0][:22], r[2], r[3]) for r in rs if r[3] - r[2] > 0.8)))
