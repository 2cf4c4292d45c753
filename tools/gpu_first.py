import ctypes, numpy as np, time, sys
sys.path.insert(0, ".")
from libzkp_amd import _native
L=_native.lib()
orc=ctypes.CDLL("oracle/_build/libzkp_oracle.so"); orc.zkp_oracle_init()
u64=ctypes.c_uint64
P=lambda a: a.ctypes.data_as(ctypes.c_void_p)
t0=time.time(); rc=L.zkp_hip_init(0); print("init rc",rc,_native.last_error(),"%.2fs"%(time.time()-t0))
def batch(n,seed):
    rng=np.random.default_rng(seed)
    v=rng.integers(0,2**32,n,dtype=np.uint64); mn=np.zeros(n,dtype=np.uint64); mx=np.full(n,2**32,dtype=np.uint64)
    seeds=rng.integers(0,256,32*n,dtype=np.uint8)
    return v,mn,mx,seeds
for n in (1,5,300):
    v,mn,mx,seeds=batch(n,n)
    out=np.zeros((n,1478),dtype=np.uint8); lens=np.zeros(n,dtype=np.uint32); st=np.zeros(n,dtype=np.int32)
    t0=time.time()
    rc=L.zkp_hip_prove_range_batch(u64(n),P(v),P(mn),P(mx),64,P(seeds),P(out),u64(1478),P(lens),P(st))
    t1=time.time()
    o2=np.zeros((n,1478),dtype=np.uint8); l2=np.zeros(n,dtype=np.uint32); s2=np.zeros(n,dtype=np.int32)
    orc.zkp_oracle_prove_range_batch(u64(n),P(v),P(mn),P(mx),64,P(seeds),P(o2),u64(1478),P(l2),P(s2),16)
    eq=(out==o2).all()
    print("n",n,"rc",rc,_native.last_error() if rc<0 else "","gpu %.1f ms"%((t1-t0)*1e3),"equal",eq, "lens ok",(lens==1478).all())
    if not eq:
        for i in range(min(n,3)):
            d=np.nonzero(out[i]!=o2[i])[0]; print(" op",i,"ndiff",len(d),d[:8])
for n in (4096,4096):
    v,mn,mx,seeds=batch(n,1)
    out=np.zeros((n,1478),dtype=np.uint8); lens=np.zeros(n,dtype=np.uint32); st=np.zeros(n,dtype=np.int32)
    t0=time.time()
    rc=L.zkp_hip_prove_range_batch(u64(n),P(v),P(mn),P(mx),64,P(seeds),P(out),u64(1478),P(lens),P(st))
    t1=time.time()
    print("n",n,"rc",rc,"host-buffer call %.1f ms -> %.0f proofs/s"%((t1-t0)*1e3,n/(t1-t0)))
ok=np.zeros(n,dtype=np.uint8)
t0=time.time()
allok=orc.zkp_oracle_verify_range_batch(u64(n),P(out),u64(1478),P(lens),P(mn),P(mx),P(ok),16)
print("oracle verifier accepts all 4096:",allok,"%.1fs"%(time.time()-t0))
