import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, ROOT)
import torch, ntt_cuda_amd as ntt
from bench import Q60, PSI60
n=32768
ctx = ntt.NTTContext(n, Q60, PSI60)
dev=torch.device("cuda",0)
def run(nstreams, per, reps=100):
    bufs=[torch.empty((per,n),dtype=torch.int64,device=dev) for _ in range(nstreams)]
    for i,b in enumerate(bufs): ctx.synth_splitmix(b, per, 1+i*per)
    streams=[torch.cuda.Stream() for _ in range(nstreams)]
    torch.cuda.synchronize()
    def step():
        for s,b in zip(streams,bufs):
            with torch.cuda.stream(s):
                ctx.forward_batch(b, per, stream=s); ctx.inverse_batch(b, per, stream=s)
    for _ in range(60): step()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(reps): step()
    torch.cuda.synchronize(); el=time.perf_counter()-t0
    return nstreams*per*reps/el
for cfg in [(1,1024),(2,512),(4,256),(2,1024),(1,2048),(4,512),(1,512),(3,768)]:
    r=[run(*cfg) for _ in range(2)]
    print("streams=%d polys/stream=%4d  => %.3f %.3f M pairs/s" % (cfg[0],cfg[1],r[0]/1e6,r[1]/1e6))
