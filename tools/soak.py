"""Randomised soak on one MI355X (development aid): many small and medium batches of every Bulletproofs framing with edge values,
GPU bytes against oracle/c, GPU verifier over everything; Groth16 / STARK batches through the GPU verifiers.
Usage: soak.py [seconds]   Exit code 1 on the first mismatch."""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import libzkp_amd as z
from libzkp_amd import _native, api
import __graft_entry__ as ge
ge.build_oracle()
oc = ctypes.CDLL(ge.ORACLE_LIB)
L = _native.lib()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
U64 = ctypes.c_uint64
_native.check(L.zkp_hip_init(0), "init")
for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
    api.install_proving_key(kind, open(os.path.join(ROOT, "tests", "golden", name), "rb").read())
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(time.time()))
EDGE = np.array([0, 1, 2, 2**8 - 1, 2**8, 2**16 - 1, 2**16, 2**32 - 1, 2**32, 2**63 - 1, 2**63, 2**64 - 2, 2**64 - 1], dtype=np.uint64)
t_end = time.time() + budget
it = 0
counts = {"range": 0, "threshold": 0, "consistency": 0, "equality": 0, "membership": 0, "improvement": 0}
while time.time() < t_end:
    it += 1
    n = int(rng.choice([1, 2, 3, 7, 64, 65, 100, 257, 600]))
    bits = int(rng.choice([8, 16, 32, 64, 64, 64]))
    cap = np.uint64(2**bits - 1)
    # ---- range: mix of edge and random bounds, widths within the capacity
    a = rng.choice(EDGE, n); b = rng.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, n, dtype=np.uint64)
    mn = np.where(rng.random(n) < 0.5, a, b)
    room = np.uint64(2**64 - 1) - mn
    span = np.minimum(np.minimum(rng.integers(0, 2**63, n, dtype=np.uint64) >> rng.integers(0, 63, n, dtype=np.uint64), cap), room)
    mx = mn + span
    off = np.where(rng.random(n) < 0.3, np.where(rng.random(n) < 0.5, np.uint64(0), span), rng.integers(0, 2**63, n, dtype=np.uint64) % (span + np.uint64(1)))
    v = mn + np.minimum(off, span)
    seeds = np.frombuffer(rng.bytes(32 * n), dtype=np.uint8).copy()
    out = np.zeros((n, 1478), dtype=np.uint8); ln = np.zeros(n, dtype=np.uint32); st = np.zeros(n, dtype=np.int32)
    o2 = np.zeros((n, 1478), dtype=np.uint8); l2 = np.zeros(n, dtype=np.uint32); s2 = np.zeros(n, dtype=np.int32)
    rc = L.zkp_hip_prove_range_batch(n, P(v), P(mn), P(mx), bits, P(seeds), P(out), 1478, P(ln), P(st))
    rc2 = oc.zkp_oracle_prove_range_batch(U64(n), P(v), P(mn), P(mx), bits, P(seeds), P(o2), U64(1478), P(l2), P(s2), 16)
    if rc != 0 or rc2 != 0 or not (out == o2).all() or not (ln == l2).all():
        print("RANGE MISMATCH it", it, "n", n, "bits", bits, rc, rc2); sys.exit(1)
    ok = np.zeros(n, dtype=np.uint8)
    L.zkp_hip_verify_range_batch(n, P(out), 1478, P(ln), P(mn), P(mx), P(ok))
    if not (ok == 1).all():
        print("RANGE VERIFY FAIL it", it); sys.exit(1)
    counts["range"] += n
    # one to three DISTINCT flipped bits per envelope (two flips of the same bit would hand back the original envelope, which round 2's
    # version of this loop counted as "tampered and accepted"): GPU verdicts must equal the oracle's verdicts, and an envelope that
    # differs from the original must be rejected by both -- both verifiers are ours, so agreement alone would not show a malleable byte
    t = out.copy()
    nbits = ln.astype(np.int64) * 8
    k = int(rng.integers(1, 4))
    chosen = np.full((n, k), -1, dtype=np.int64)
    for j in range(k):
        b = rng.integers(0, 2**62, n) % nbits
        for _ in range(k):                                            # step off bits already flipped in this envelope
            clash = (chosen[:, :j] == b[:, None]).any(axis=1)
            b = np.where(clash, (b + 1) % nbits, b)
        chosen[:, j] = b
        t[np.arange(n), b >> 3] ^= (1 << (b & 7)).astype(np.uint8)
    assert all((t[i, :ln[i]] != out[i, :ln[i]]).sum() >= 1 for i in range(n))
    want = np.zeros(n, dtype=np.uint8)
    oc.zkp_oracle_verify_range_batch(U64(n), P(t), U64(1478), P(ln), P(mn), P(mx), P(want), 16)
    L.zkp_hip_verify_range_batch(n, P(t), 1478, P(ln), P(mn), P(mx), P(ok))
    if not (ok == want).all():
        print("TAMPER VERDICT MISMATCH it", it, "n", n, "bits", bits, np.nonzero(ok != want)[0][:5]); sys.exit(1)
    if want.any() or ok.any():
        i = int(np.nonzero(want | ok)[0][0])
        print("TAMPERED ENVELOPE ACCEPTED it", it, "op", i, "bits", bits, "flipped bit positions", chosen[i].tolist(), "oracle", int(want[i]), "gpu", int(ok[i]))
        print("original:", out[i, :ln[i]].tobytes().hex()); print("tampered:", t[i, :ln[i]].tobytes().hex()); sys.exit(1)
    counts["tampered"] = counts.get("tampered", 0) + n
    # ---- threshold / consistency through the Python mirror against the oracle's single-proof entry points
    m = min(n, 40)
    lists = [[int(x) for x in rng.integers(0, 2**20, int(rng.integers(1, 6)))] for _ in range(m)]
    thr = [int(rng.integers(0, sum(l) + 1)) for l in lists]
    sd = rng.bytes(32 * m)
    tp = z.prove_threshold_batch(lists, thr, seeds=sd)
    buf = ctypes.create_string_buffer(8192); ol = ctypes.c_uint32()
    for i in range(m):
        vals = (ctypes.c_uint64 * len(lists[i]))(*lists[i])
        if oc.zkp_oracle_prove_threshold(vals, len(lists[i]), U64(thr[i]), 64, sd[32 * i: 32 * i + 32], buf, 8192, ctypes.byref(ol)) != 0 or buf.raw[: ol.value] != tp[i]:
            print("THRESHOLD MISMATCH it", it, i); sys.exit(1)
    if not all(z.verify_threshold_batch(tp, thr)):
        print("THRESHOLD VERIFY FAIL"); sys.exit(1)
    counts["threshold"] += m
    data = [sorted(int(x) for x in rng.choice(EDGE, int(rng.integers(1, 6)))) if rng.random() < 0.3 else sorted(int(x) for x in rng.integers(0, 2**40, int(rng.integers(1, 9)))) for _ in range(m)]
    cp = z.prove_consistency_batch(data, seeds=sd)
    for i in range(m):
        vals = (ctypes.c_uint64 * len(data[i]))(*data[i])
        if oc.zkp_oracle_prove_consistency(vals, len(data[i]), sd[32 * i: 32 * i + 32], buf, 8192, ctypes.byref(ol)) != 0 or buf.raw[: ol.value] != cp[i]:
            print("CONSISTENCY MISMATCH it", it, i); sys.exit(1)
    if not all(z.verify_consistency_batch(cp)):
        print("CONSISTENCY VERIFY FAIL"); sys.exit(1)
    counts["consistency"] += m
    # ---- Groth16 and STARK: prove, then the GPU verifiers must accept (their verdicts are pinned to the oracle's in the tests)
    if it % 3 == 0:
        k = int(rng.choice([1, 3, 65, 130]))
        vals = [int(x) for x in rng.choice(EDGE, k)] if it % 2 else [int(x) for x in rng.integers(0, 2**63, k)]
        ep = z.prove_equality_batch(vals, vals)
        if not all(api._verify_snark_envelopes(0, ep)):
            print("EQUALITY VERIFY FAIL it", it); sys.exit(1)
        counts["equality"] += k
        # tampered Groth16 envelopes: one to three flipped bits, none of them a sign flag of an uncompressed point (ark ignores those):
        # the pairing check on the Fq2 machine must reject every one
        sign_bits = {(73, 7), (201, 7), (265, 7)}
        bad = []
        for e in ep:
            t = bytearray(e)
            for _ in range(int(rng.integers(1, 4))):
                while True:
                    pos, bit = int(rng.integers(0, 298)), int(rng.integers(0, 8))
                    if (pos, bit) not in sign_bits: break
                t[pos] ^= 1 << bit
            if bytes(t) != e: bad.append(bytes(t))
        acc = api._verify_snark_envelopes(0, bad)
        if any(acc):
            i = acc.index(True)
            print("TAMPERED EQUALITY ENVELOPE ACCEPTED it", it, "\noriginal:", ep[min(i, len(ep) - 1)].hex(), "\ntampered:", bad[i].hex()); sys.exit(1)
        counts["equality_tampered"] = counts.get("equality_tampered", 0) + len(bad)
        sets = [[int(x) for x in rng.choice(2**40, int(rng.integers(1, 65)), replace=False)] for _ in range(min(k, 20))]
        mp = z.prove_membership_batch([s[int(rng.integers(0, len(s)))] for s in sets], sets)
        if not all(z.verify_membership_batch(mp, sets)):
            print("MEMBERSHIP VERIFY FAIL it", it); sys.exit(1)
        counts["membership"] += len(sets)
        olds = [int(x) for x in rng.choice(EDGE[:-1], k)]
        news = [min(2**64 - 1, o + 1 + int(rng.integers(0, 2**30))) for o in olds]
        ip = z.prove_improvement_batch(olds, news)
        if not all(z.verify_improvement_batch(ip, olds)):
            print("IMPROVEMENT VERIFY FAIL it", it); sys.exit(1)
        counts["improvement"] += k
print("soak ok: %d iterations, proofs checked: %s" % (it, counts))
