import sys, numpy as np
sys.path.insert(0,'ntt-cuda_amd'); sys.path.insert(0,'oracle'); sys.path.insert(0,'tests')
import torch, ntt_cuda_amd as ntt, oracle_py as o
from test_ntt30 import PARAMS30
dev=torch.device('cuda',0)
for n in (2048, 32768):
    q,psi,_,_,bits=PARAMS30[n]
    prm=o.Params30(n,q,psi)
    rng=np.random.default_rng(1)
    a=rng.integers(0,q,size=(2,n),dtype=np.uint32)
    d=torch.from_numpy(a.view(np.int32)).to(dev)
    tabi=torch.from_numpy(prm.psiinv_tab.view(np.int32)).to(dev)
    ntt.inverse30(d,n,q,prm.mu,bits,tabi,2)
    torch.cuda.synchronize()
    got=d.cpu().numpy().view(np.uint32); want=o.inverse30(a,prm)
    bad=np.nonzero(got!=want)
    print(n,"mismatches",len(bad[0]),"of",got.size, "first idx", bad[1][:20])
    # repeat the call: does a second call differ?
    d2=torch.from_numpy(a.view(np.int32)).to(dev)
    ntt.inverse30(d2,n,q,prm.mu,bits,tabi,2); torch.cuda.synchronize()
    got2=d2.cpu().numpy().view(np.uint32)
    print("  second call mismatches", int((got2!=want).sum()))
