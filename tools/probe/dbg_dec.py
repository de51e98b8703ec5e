import sys, numpy as np
sys.path[:0]=['ntt-cuda_amd','oracle','tests']
import ntt_cuda_amd as native, oracle_py as oracle, params as P, torch
from ntt_cuda_amd import bfv
oracle.build()
n,t=4096,1024
qs=[q for q,_ in P.Q62_N4096]; psis=[w for _,w in P.Q62_N4096]; R=len(qs)
for gamma in (P.GAMMA61, P.GAMMA40):
    ctx=bfv.BFVContext(n,qs,psis,t,gamma)
    rng=np.random.default_rng(2024)
    c=np.stack([np.stack([rng.integers(0,q,size=n,dtype=np.uint64) for q in qs]) for _ in range(2)])
    sk=oracle.bfv_sample(qs,n,9)["uniform"]
    w_c=c.reshape(-1).copy(); w_m=np.empty(n,np.uint64)
    stage=np.empty(3*(R-1)*n,np.uint64)
    oracle.lib().orc_bfv_decrypt(oracle._p(w_c), oracle._p(np.ascontiguousarray(sk.reshape(-1)[:(R-1)*n])), oracle._p(np.array(qs,np.uint64)), oracle._p(np.array(psis,np.uint64)), R, n, t, gamma, oracle._p(w_m), oracle._p(stage))
    d_c=native.to_device(c); d_sk=native.to_device(sk)
    m=ctx.decrypt(d_c,d_sk); torch.cuda.synchronize()
    g=native.to_host(d_c).reshape(2,R,n); w=w_c.reshape(2,R,n)
    print('gamma bits',gamma.bit_length(),'diff per poly', [[int((g[h,i]!=w[h,i]).sum()) for i in range(R)] for h in range(2)], 'm diff', int((native.to_host(m)!=w_m).sum()))
    # NTT section alone
    nctx=native.NTTContext(n,qs,psis)
    c1=native.to_device(c[1,:R-1].copy()); 
    nctx.forward_batch(c1,R-1,division=R)
    f=native.to_host(c1).reshape(R-1,n)
    print('  forward stage diff', int((f.reshape(-1)!=stage[:(R-1)*n]).sum()))
# column 0 by hand from the GPU's own c1 (after the scale step, which agrees with the oracle)
gamma=P.GAMMA40
ctx=bfv.BFVContext(n,qs,psis,t,gamma)
Mx=(1<<64)-1
def barrett(a,b,q,mu,k):
    P_=a*b; x1=(P_>>(k-2))&Mx; s=((x1*mu)>>(k+2))&Mx; r_=(P_-s*q)&Mx
    return r_-q if r_>=q else r_
k=gamma.bit_length(); mu=(1<<(2*k))//gamma
r=R-1
c1s=[int(w[1,i,0]) for i in range(r)]
bcm=[]
for j in range(r):
    tmp=1
    for kk in range(r):
        if kk!=j: tmp=tmp*qs[kk]%gamma
    bcm.append(tmp)
acc=0
for v,b in zip(c1s,bcm): acc=((acc+barrett(v,b,gamma,mu,k))&Mx)%gamma
mult=1
for i in range(r): mult=mult*qs[i]%gamma
neg=gamma-pow(mult,gamma-2,gamma)
x1=barrett(acc,neg,gamma,mu,k)
print('by hand acc',acc,'x1',x1,'oracle x1',int(w[0,1,0]),'gpu x1',int(g[0,1,0]))
print('constants', ctx.constants() if hasattr(ctx,'constants') else None)
