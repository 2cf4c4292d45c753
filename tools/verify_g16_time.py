"""Groth16 verification timing on one MI355X (development aid): proves n equality envelopes (and n // 4 membership envelopes with
16-element sets), verifies each batch five times through the C ABI and prints the best wall time.  Under
`rocprofv3 --kernel-trace --stats -- python3 tools/verify_g16_time.py` the kernel table shows where the time goes.
Usage: verify_g16_time.py [n] [--sweep]   (--sweep: also equality batches of 1024 / 16384 / 65536 envelopes, cycled copies of the n proved)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libzkp_amd as z
from libzkp_amd import _native, api
L = _native.lib()
_native.check(L.zkp_hip_init(0), "init")
for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
    api.install_proving_key(kind, open(os.path.join(ROOT, "tests", "golden", name), "rb").read())
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 4096
rng = np.random.default_rng(11)


def best(f, reps=5):
    f()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = f(); ts.append(time.perf_counter() - t0)
    return min(ts), r


vals = [int(x) for x in rng.integers(0, 2**63, n)]
ep = z.prove_equality_batch(vals, vals)
dt, ok = best(lambda: api._verify_snark_envelopes(0, ep))
assert all(ok)
bad = [bytes(e[:40]) + bytes([e[40] ^ 1]) + bytes(e[41:]) for e in ep[:64]]
assert not any(api._verify_snark_envelopes(0, bad))
out = {"equality": {"n": n, "ms": round(dt * 1e3, 2), "envelopes_per_s": round(n / dt)}}
# the C ABI alone (the Python mirror's per-envelope marshalling left out): one strided buffer in, verdict bytes out
import ctypes
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
buf = np.zeros((n, 298), dtype=np.uint8); lens = np.full(n, 298, dtype=np.uint32); okb = np.zeros(n, dtype=np.uint8)
for i, e in enumerate(ep): buf[i] = np.frombuffer(e, dtype=np.uint8)
dt, _ = best(lambda: _native.check(L.zkp_hip_verify_equality_batch(n, P(buf), 298, P(lens), P(okb)), "verify"))
assert okb.all()
out["equality"]["c_abi_ms"] = round(dt * 1e3, 2); out["equality"]["c_abi_envelopes_per_s"] = round(n / dt)
m = max(1, n // 4)
sets = [[int(x) for x in rng.choice(2**40, 16, replace=False)] for _ in range(m)]
mp = z.prove_membership_batch([s[3] for s in sets], sets)
dt, ok = best(lambda: z.verify_membership_batch(mp, sets))
assert all(ok)
out["membership"] = {"n": m, "set": 16, "ms": round(dt * 1e3, 2), "envelopes_per_s": round(m / dt)}
ml = len(mp[0]); mbuf = np.zeros((m, ml), dtype=np.uint8); mlens = np.full(m, ml, dtype=np.uint32); mok = np.zeros(m, dtype=np.uint8)
for i, e in enumerate(mp): mbuf[i] = np.frombuffer(e, dtype=np.uint8)
dt, _ = best(lambda: _native.check(L.zkp_hip_verify_membership_batch(m, P(mbuf), ml, P(mlens), P(mok)), "verify"))
assert mok.all()
out["membership"]["c_abi_ms"] = round(dt * 1e3, 2); out["membership"]["c_abi_envelopes_per_s"] = round(m / dt)
if "--sweep" in sys.argv:
    out["membership_sweep"] = {}
    for m2 in (4096, 16384):
        big = np.ascontiguousarray(mbuf[np.arange(m2) % m]); bl = np.full(m2, ml, dtype=np.uint32); bok = np.zeros(m2, dtype=np.uint8)
        dt, _ = best(lambda: _native.check(L.zkp_hip_verify_membership_batch(m2, P(big), ml, P(bl), P(bok)), "verify"), reps=3)
        assert bok.all()
        out["membership_sweep"][str(m2)] = {"c_abi_ms": round(dt * 1e3, 2), "c_abi_envelopes_per_s": round(m2 / dt)}
if "--sweep" in sys.argv:
    out["equality_sweep"] = {}
    for m2 in (1024, 8192, 16384, 65536):
        batch = [ep[i % n] for i in range(m2)]
        dt, ok = best(lambda: api._verify_snark_envelopes(0, batch), reps=3)
        assert all(ok)
        out["equality_sweep"][str(m2)] = {"ms": round(dt * 1e3, 2), "envelopes_per_s": round(m2 / dt)}
        big = np.ascontiguousarray(buf[np.arange(m2) % n]); bl = np.full(m2, 298, dtype=np.uint32); bok = np.zeros(m2, dtype=np.uint8)
        dt, _ = best(lambda: _native.check(L.zkp_hip_verify_equality_batch(m2, P(big), 298, P(bl), P(bok)), "verify"), reps=3)
        assert bok.all()
        out["equality_sweep"][str(m2)].update({"c_abi_ms": round(dt * 1e3, 2), "c_abi_envelopes_per_s": round(m2 / dt)})
        big[m2 // 3, 266:298] = big[m2 // 3 + 1, 266:298]       # one envelope under its neighbour's commitment: the batch check fails, every envelope is verified again
        dt, _ = best(lambda: _native.check(L.zkp_hip_verify_equality_batch(m2, P(big), 298, P(bl), P(bok)), "verify"), reps=2)
        assert bok.sum() == m2 - 1 and not bok[m2 // 3]
        out["equality_sweep"][str(m2)]["c_abi_ms_with_one_tampered"] = round(dt * 1e3, 2)
print(json.dumps(out))
