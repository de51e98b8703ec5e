import os, sys
sys.path.insert(0,'ntt-cuda_amd'); sys.path.insert(0,'.')
import torch
import ntt_cuda_amd as ntt
from bench import Q60, PSI60, synth
dev=torch.device('cuda',0)
ctx=ntt.NTTContext(32768,Q60,PSI60)
for num in (32,64,128,192,256,384,512):
    a=synth(torch,num,32768,Q60,dev,1); b=synth(torch,num,32768,Q60,dev,2)
    for _ in range(20): ctx.polymul_batch(a,b,num)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ctx.polymul_batch(a,b,num)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/100*1e3
    print("num=%4d polymul %7.1f us => %.3f M/s"%(num,us,num/us))
