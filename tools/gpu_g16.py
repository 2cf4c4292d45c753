import ctypes, json, sys, time
import numpy as np
sys.path.insert(0, ".")
from libzkp_amd import _native
from oracle.py import groth16 as g
L=_native.lib()
P=lambda a: a.ctypes.data_as(ctypes.c_void_p)
vec=json.load(open("tests/golden/groth16_vectors.json"))
ss=bytes.fromhex(vec["setup_seed"])
vals=np.array([int(k) for k in vec["mimc"]],dtype=np.uint64)
out=np.zeros((len(vals),32),dtype=np.uint8)
rc=L.zkp_hip_snark_commit_value_batch(len(vals),P(vals),P(out)); print("mimc rc",rc,_native.last_error() if rc else "")
print("mimc ok", all(out[i].tobytes().hex()==vec["mimc"][str(int(v))] for i,v in enumerate(vals)))
pk=open("tests/golden/equality_mimc_pk.bin","rb").read()
t0=time.time(); rc=L.zkp_hip_groth16_load_key(0,pk,len(pk)); print("load eq key rc",rc,_native.last_error() if rc else "","%.2fs"%(time.time()-t0))
n=5
v=np.array([42,0,2**64-1,123456789,7],dtype=np.uint64); v2=v.copy(); v2[4]=8
seeds=(np.arange(32*n,dtype=np.uint32)*3+1).astype(np.uint8)
o=np.zeros((n,298),dtype=np.uint8); ln=np.zeros(n,dtype=np.uint32); st=np.zeros(n,dtype=np.int32)
t0=time.time(); rc=L.zkp_hip_prove_equality_batch(n,P(v),P(v2),P(seeds),P(o),298,P(ln),P(st)); print("prove eq rc",rc,_native.last_error() if rc<0 else "",list(ln),list(st),"%.2fs"%(time.time()-t0))
key=g.equality_key(ss)
for i in range(4):
    val=int(v[i]); sd=seeds[32*i:32*i+32].tobytes()
    cm=g.commit_value_snark(val)
    cs=g.equality_circuit(val,val,int.from_bytes(cm,'little'))
    ref=g.envelope(2,g.prove_with_trapdoor(key,cs,g.draw_fr(sd,0x47313600,0),g.draw_fr(sd,0x47313600,1)),cm)
    got=o[i].tobytes()
    print(" op",i,"equal",got==ref, "" if got==ref else [k for k in range(298) if got[k]!=ref[k]][:6])
print("pairing verify op0:", g.verify_equality_with_commitment(o[0].tobytes(), g.commit_value_snark(42), ss))
pk=open("tests/golden/membership_mimc_pk.bin","rb").read()
t0=time.time(); rc=L.zkp_hip_groth16_load_key(1,pk,len(pk)); print("load mem key rc",rc,_native.last_error() if rc else "","%.2fs"%(time.time()-t0))
sets=[[10,20,25,30,40],[5],[7,7,9]]; mv=np.array([25,5,9],dtype=np.uint64)
flat=np.array([x for s in sets for x in s],dtype=np.uint64); cnt=np.array([len(s) for s in sets],dtype=np.uint32)
stride=10+4+8*5+256+32
o=np.zeros((3,stride),dtype=np.uint8); ln=np.zeros(3,dtype=np.uint32); st=np.zeros(3,dtype=np.int32)
rc=L.zkp_hip_prove_membership_batch(3,P(mv),P(flat),P(cnt),P(seeds),P(o),stride,P(ln),P(st)); print("prove mem rc",rc,_native.last_error() if rc<0 else "",list(ln),list(st))
mkey=g.membership_key(ss)
for i in range(3):
    val=int(mv[i]); sd=seeds[32*i:32*i+32].tobytes(); cm=g.commit_value_snark(val)
    sel,sv,ir=g.membership_inputs(val,sets[i])
    cs=g.membership_circuit(val,sel,sv,ir,int.from_bytes(cm,'little'))
    pr=g.prove_with_trapdoor(mkey,cs,g.draw_fr(sd,0x47313600,0),g.draw_fr(sd,0x47313600,1))
    payload=len(sets[i]).to_bytes(4,'little')+b"".join(x.to_bytes(8,'little') for x in sets[i])+pr
    ref=g.envelope(4,payload,cm); got=o[i,:ln[i]].tobytes()
    print(" mem op",i,"equal",got==ref)
print("pairing verify mem op0:", g.verify_membership(o[0,:ln[0]].tobytes(), sets[0], ss))
