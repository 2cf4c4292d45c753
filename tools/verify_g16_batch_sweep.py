"""Groth16 equality verification through the C ABI at several batch sizes, with the batch check (g16_rlc.h: one pairing check per call) and with
the per-envelope check: where the default threshold (ZKP_HIP_G16_BATCH_VERIFY_MIN) belongs and what the batch check buys above it.  Each mode
runs in a child process (the switches are read once).  Prints one JSON object (profiles/r04_verify_g16_batch.json).
Usage: verify_g16_batch_sweep.py [sizes ...]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [int(a) for a in sys.argv[1:] if a.isdigit()] or [2048, 4096, 8192, 10240, 12288, 16384, 32768, 65536]

if "--child" in sys.argv:
    import ctypes
    import numpy as np
    sys.path.insert(0, ROOT)
    import libzkp_amd as z
    from libzkp_amd import _native, api
    L = _native.lib()
    _native.check(L.zkp_hip_init(0), "init")
    for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
        api.install_proving_key(kind, open(os.path.join(ROOT, "tests", "golden", name), "rb").read())
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rng = np.random.default_rng(3)
    n0 = 4096
    vals = [int(x) for x in rng.integers(0, 2**63, n0)]
    ep = z.prove_equality_batch(vals, vals)
    buf = np.zeros((n0, 298), dtype=np.uint8)
    for i, e in enumerate(ep): buf[i] = np.frombuffer(e, dtype=np.uint8)
    sets = [[int(x) for x in rng.choice(2**40, 16, replace=False)] for _ in range(1024)]
    mp = z.prove_membership_batch([s[3] for s in sets], sets)
    ml = len(mp[0]); mbuf = np.zeros((1024, ml), dtype=np.uint8)
    for i, e in enumerate(mp): mbuf[i] = np.frombuffer(e, dtype=np.uint8)
    out = {"equality": {}, "membership_16": {}}
    for name, src, width, fn in (("equality", buf, 298, L.zkp_hip_verify_equality_batch), ("membership_16", mbuf, ml, L.zkp_hip_verify_membership_batch)):
        for m in SIZES:
            big = np.ascontiguousarray(src[np.arange(m) % src.shape[0]]); bl = np.full(m, width, dtype=np.uint32); ok = np.zeros(m, dtype=np.uint8)
            ts = []
            for _ in range(6):
                t0 = time.perf_counter(); _native.check(fn(m, P(big), width, P(bl), P(ok)), "verify"); ts.append(time.perf_counter() - t0)
            assert ok.all()
            out[name][str(m)] = round(min(ts[1:]) * 1e3, 2)
    print(json.dumps(out))
    sys.exit(0)

res = {}
for mode, env in (("batch_check", {"ZKP_HIP_G16_BATCH_VERIFY_MIN": "1"}), ("per_envelope", {"ZKP_HIP_NO_BATCH_VERIFY": "1"})):
    o = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + [str(s) for s in SIZES], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    if o.returncode != 0:
        sys.stderr.write(o.stderr[-3000:]); sys.exit(1)
    res[mode] = json.loads(o.stdout.strip().splitlines()[-1])
rows = []
for kind in ("equality", "membership_16"):
    for s in SIZES:
        b, p = res["batch_check"][kind][str(s)], res["per_envelope"][kind][str(s)]
        rows.append({"circuit": kind, "envelopes": s, "batch_check_ms": b, "per_envelope_ms": p, "batch_check_envelopes_per_s": round(s / b * 1e3), "per_envelope_envelopes_per_s": round(s / p * 1e3)})
print(json.dumps({"tool": "tools/verify_g16_batch_sweep.py", "entry": "zkp_hip_verify_equality_batch / zkp_hip_verify_membership_batch", "timing": "host wall clock, best of 5, host buffers in, verdict bytes out; all envelopes valid", "rows": rows}))
