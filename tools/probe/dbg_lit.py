import sys, numpy as np
sys.path[:0]=['ntt-cuda_amd','oracle','tests']
import ntt_cuda_amd as native, oracle_py as oracle, params as P, torch
oracle.build()
n=2048
def rt(q, n):
    for x in range(2,1000):
        w=pow(x,(q-1)//(2*n),q)
        if pow(w,n,q)==q-1: return w
cases={'in36+ex60':[P.INEXACT_PRIMES[36],P.EXACT_NEIGHBOURS[60]], 'in36+ex36':[P.INEXACT_PRIMES[36],P.EXACT_NEIGHBOURS[36]],
       'in36+Q60':[P.INEXACT_PRIMES[36],(P.Q60[0],{n:rt(P.Q60[0],n)})], 'in36+gen60':[P.INEXACT_PRIMES[36],(P.GENERAL_PRIMES[60][0],{n:rt(P.GENERAL_PRIMES[60][0],n)})],
       'in36+edge62':[P.INEXACT_PRIMES[36],(P.EDGE_PRIMES[62][0],{n:rt(P.EDGE_PRIMES[62][0],n)})]}
for name,sel in cases.items():
    qs=[q for q,_ in sel]; psis=[r[n] for _,r in sel]
    ctx=native.NTTContext(n,qs,psis); prm=oracle.Params(n,qs,psis)
    num=4
    a=oracle.synth_batch(n,num,qs,3).reshape(num,n)
    want=oracle.forward_batch(a.copy(),prm).reshape(num,n)
    d=native.to_device(a); ctx.forward_batch(d,num); torch.cuda.synchronize()
    got=native.to_host(d).reshape(num,n)
    print(name, ctx.kernel_class, ctx.literal_routing, [int((got[y]!=want[y]).sum()) for y in range(num)], [native.barrett_is_exact(q) for q in qs])
    y=1
    if (got[y]!=want[y]).any():
        i=np.nonzero(got[y]!=want[y])[0][:4]; print('   idx',i, got[y][i], want[y][i], (got[y][i].astype(object)-want[y][i].astype(object)))
