"""A/B debugging aid: prove a few equality / membership envelopes with fixed seeds, verify them, save them; a second run
(other library build via ZKP_HIP_LIB) compares its own bytes with the saved ones and verifies the saved ones."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libzkp_amd import _native
L = _native.lib()
P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
tag = sys.argv[1]
for kind, name in ((0, "equality_mimc_pk.bin"), (1, "membership_mimc_pk.bin")):
    pk = open(os.path.join(ROOT, "tests/golden", name), "rb").read()
    assert L.zkp_hip_groth16_load_key(kind, pk, len(pk)) == 0
n = 8
v = np.arange(100, 100 + n, dtype=np.uint64); seeds = np.arange(32 * n, dtype=np.uint8)
o = np.zeros((n, 298), dtype=np.uint8); ln = np.zeros(n, dtype=np.uint32); st = np.zeros(n, dtype=np.int32)
assert L.zkp_hip_prove_equality_batch(n, P(v), P(v), P(seeds), P(o), 298, P(ln), P(st)) == 0
ok = np.zeros(n, dtype=np.uint8)
L.zkp_hip_verify_equality_batch(n, P(o), 298, P(ln), P(ok))
print(tag, "own equality proofs verify:", ok.tolist())
path = os.path.join(ROOT, "gpurun_out", "r2g", "eq_proofs.npy")
if os.path.exists(path):
    other = np.load(path)
    print(tag, "bytes equal to the other build's:", bool((other == o).all()))
    L.zkp_hip_verify_equality_batch(n, P(other), 298, P(ln), P(ok))
    print(tag, "other build's proofs verify here:", ok.tolist())
else:
    np.save(path, o)
sets = np.arange(16 * n, dtype=np.uint64).reshape(n, 16) * 3 + 7; mv = sets[np.arange(n), np.arange(n) % 16].copy(); cnt = np.full(n, 16, dtype=np.uint32)
stride = 10 + 4 + 128 + 256 + 32
mo = np.zeros((n, stride), dtype=np.uint8)
assert L.zkp_hip_prove_membership_batch(n, P(mv), P(sets.ravel().copy()), P(cnt), P(seeds), P(mo), stride, P(ln), P(st)) == 0
L.zkp_hip_verify_membership_batch(n, P(mo), stride, P(ln), P(ok))
print(tag, "own membership proofs verify:", ok.tolist())
