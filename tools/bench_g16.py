import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from libzkp_amd import _native
L=_native.lib()
P=lambda a: a.ctypes.data_as(ctypes.c_void_p)
kind=sys.argv[1] if len(sys.argv)>1 else "equality"
n=int(sys.argv[2]) if len(sys.argv)>2 else 4096
if len(sys.argv)>3: L.zkp_hip_set_window_budget(10000+int(sys.argv[3]))   # force the MSM chunk count
rng=np.random.default_rng(2)
seeds=rng.integers(0,256,32*n,dtype=np.uint8)
if kind=="equality":
    pk=open(os.path.join(ROOT,"tests/golden/equality_mimc_pk.bin"),"rb").read()
    t0=time.time(); assert L.zkp_hip_groth16_load_key(0,pk,len(pk))==0; print("key load %.2fs"%(time.time()-t0))
    v=rng.integers(0,2**63,n,dtype=np.uint64)
    o=np.zeros((n,298),dtype=np.uint8); ln=np.zeros(n,dtype=np.uint32); st=np.zeros(n,dtype=np.int32)
    call=lambda: L.zkp_hip_prove_equality_batch(n,P(v),P(v),P(seeds),P(o),298,P(ln),P(st))
else:
    pk=open(os.path.join(ROOT,"tests/golden/membership_mimc_pk.bin"),"rb").read()
    t0=time.time(); assert L.zkp_hip_groth16_load_key(1,pk,len(pk))==0; print("key load %.2fs"%(time.time()-t0))
    sets=rng.integers(0,2**32,(n,16),dtype=np.uint64); v=sets[np.arange(n),np.arange(n)%16].copy()
    cnt=np.full(n,16,dtype=np.uint32); flat=sets.ravel().copy()
    stride=10+4+8*16+256+32
    o=np.zeros((n,stride),dtype=np.uint8); ln=np.zeros(n,dtype=np.uint32); st=np.zeros(n,dtype=np.int32)
    call=lambda: L.zkp_hip_prove_membership_batch(n,P(v),P(flat),P(cnt),P(seeds),P(o),stride,P(ln),P(st))
for it in range(4):
    t0=time.time(); rc=call(); dt=time.time()-t0
    print(kind,"n",n,"rc",rc,"%.1f ms -> %.0f proofs/s"%(dt*1e3,n/dt), "ok" if (st==0).all() else "FAIL")
if kind == "equality":
    ok = np.zeros(n, dtype=np.uint8)
    for it in range(2):
        t0 = time.time(); rc = L.zkp_hip_verify_equality_batch(n, P(o), 298, P(ln), P(ok)); dt = time.time() - t0
        print("verify equality n", n, "rc", rc, "%.1f ms -> %.0f envelopes/s" % (dt * 1e3, n / dt), "all ok" if (ok == 1).all() else "FAIL")
