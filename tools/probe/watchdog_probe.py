#!/usr/bin/env python3
"""probe: which foreign loads make a cooperating-workgroup launch give up (MI355NTT_PAIR_WATCHDOG_MS=20 in the environment)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ntt_cuda_amd as ntt, params as P
dev = torch.device("cuda", 0)
n, num = 65536, 96
qs = P.Q60[:2]
psis = [next(x for x in (pow(g, (q - 1) // (2 * n), q) for g in range(2, 2000)) if pow(x, n, q) == q - 1) for q in qs]
ctx = ntt.NTTContext(n, qs, psis)
a = torch.empty((num, n), dtype=torch.int64, device=dev); ctx.synth_splitmix(a, num, 5)
good = a.clone(); ctx.forward_batch(good, num); torch.cuda.synchronize()
cus = torch.cuda.get_device_properties(dev).multi_processor_count
print("cus", cus, "watchdog env", os.environ.get("MI355NTT_PAIR_WATCHDOG_MS"))
load, s1 = torch.cuda.Stream(), torch.cuda.Stream()
for held in (cus - 1, cus - 2, cus - 4, cus - 8, cus - 16, cus - 32, cus // 2):
    w = a.clone(); torch.cuda.synchronize()
    ctx.occupy(held, 300000, stream=load)
    time.sleep(0.02)
    t0 = time.perf_counter()
    rc = "ok"
    try:
        ctx.forward_batch(w, num, stream=s1)
    except ntt.NTTError as e:
        rc = "err %d hip %d" % (e.code, ntt.lib().mi355ntt_last_hip_error())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("held %3d CUs: call %s, %.3f s, result %s" % (held, rc, dt, "right" if torch.equal(w, good) else "WRONG"))
    # drain a pending error so the next round starts clean
    for _ in range(2):
        w2 = a.clone()
        try:
            ctx.forward_batch(w2, num, stream=s1); torch.cuda.synchronize()
            print("     follow-up call ok, result", "right" if torch.equal(w2, good) else "WRONG")
        except ntt.NTTError as e:
            print("     follow-up call reported err %d hip %d" % (e.code, ntt.lib().mi355ntt_last_hip_error()))
