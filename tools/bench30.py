import sys, time
sys.path.insert(0,'ntt-cuda_amd'); sys.path.insert(0,'oracle')
import torch, numpy as np
import ntt_cuda_amd as ntt, oracle_py as o
dev=torch.device('cuda',0)
for n,(q,psi) in {32768:(19070977,377),4096:(33538049,2386)}.items():
    prm=o.Params30(n,q,psi)
    num=(1<<27)//n   # 512 MiB of u32
    a=torch.randint(0,q,(num,n),dtype=torch.int32,device=dev)
    tab=torch.from_numpy(prm.psi_tab.view(np.int32)).to(dev); tabi=torch.from_numpy(prm.psiinv_tab.view(np.int32)).to(dev)
    for _ in range(5): ntt.forward30(a,n,q,prm.mu,prm.k,tab,num); ntt.inverse30(a,n,q,prm.mu,prm.k,tabi,num)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ntt.forward30(a,n,q,prm.mu,prm.k,tab,num)
    e1.record(); torch.cuda.synchronize(); f=e0.elapsed_time(e1)/20
    e0.record()
    for _ in range(20): ntt.inverse30(a,n,q,prm.mu,prm.k,tabi,num)
    e1.record(); torch.cuda.synchronize(); i=e0.elapsed_time(e1)/20
    gb=num*n*4*2/1e9
    print("30-bit n=%d num=%d: forward %.3f ms (%.0f GB/s alg, %.2f M/s)  inverse %.3f ms (%.0f GB/s)"%(n,num,f,gb/f*1e3,num/f/1e3,i,gb/i*1e3))
