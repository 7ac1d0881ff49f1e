import sys, os, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests")); sys.path.insert(0,os.path.join(ROOT,"oracle"))
import babyjubjub_rs_amd as bjj
from conftest import Oracle
from babyjubjub_rs_amd.workload import make_signatures, corrupt
import bjj_oracle as o
L=o.SUBORDER
ctx=bjj.Context(0); orc=Oracle()
n=1<<20
A,R,S,msg=make_signatures(ctx.mul_fixed_base, ctx.poseidon5, n)
bad=corrupt(A,R,S,msg,n)
got=ctx.eddsa_verify(A,R,S,msg)
want=(~bad).astype(np.uint8)
mis=np.nonzero(got!=want)[0]
print("mismatches:",mis.size,"of",n, "first:",mis[:10])
def short_pair_best(kappa):
    r0,t0,r1,t1 = L,0,kappa,1
    while r1 >= (1<<126):
        q=r0//r1; r0,r1 = r1, r0-q*r1; t0,t1 = t1, t0-q*t1
    if t1%2: return r1,t1,"cur"
    cands=[(r0,t0,"P")]
    if r1:
        q=r0//r1; r2,t2 = r0-q*r1, t0-q*t1
        if t2%L: cands.append((r2,t2,"N"))
    return min(cands,key=lambda c:max(c[0].bit_length(),abs(c[1]).bit_length()))
hm=ctx.poseidon5(np.concatenate([R,A,msg],axis=1))
for i in mis[:12]:
    k=int.from_bytes(hm[i].tobytes(),'little')%L
    u,v,which=short_pair_best(k)
    print(i, "got",got[i],"want",want[i],"oracle",orc.verify(A[i:i+1],R[i:i+1],S[i:i+1],msg[i:i+1])[0], "bits u",u.bit_length(),"v",abs(v).bit_length(),which, "bad" ,bad[i])
# rerun: deterministic?
got2=ctx.eddsa_verify(A,R,S,msg); print("rerun same mismatches:", np.array_equal(got,got2))
# small batch containing the mismatching items only
if mis.size:
    sub=mis[:64]
    g3=ctx.eddsa_verify(A[sub],R[sub],S[sub],msg[sub]); print("sub-batch verdicts", g3[:16], "want", want[sub][:16])
