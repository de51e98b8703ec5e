#!/usr/bin/env python3
"""rocprofv3 driver for the configs[4] pipeline (round 5): the batched BFV drivers (64 ciphertexts per call,
layout [2][64][R][n]) and the complete single-ciphertext drivers (keystream + samplers + transforms) at n = 32768 on the 4 + 1
60-bit primes of bench.py and on the reference's published 16-prime set (demo.cu:35-36).
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT/trace -- python3 tools/prof_driver_bfv.py [reps]
tools/prof_summary_bfv.py turns the kernel stats into per-kernel us, algorithmic bytes and TB/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd"))
sys.path.insert(0, ROOT)
import torch
import ntt_cuda_amd as ntt
if os.environ.get('MI355NTT_LIB'):          # (A/B against another build of the library)
    ntt.LIB_PATH = os.environ['MI355NTT_LIB']
from ntt_cuda_amd import bfv
from bench import Q60, PSI60, Q60_SPECIAL, PSI60_SPECIAL, DEMO_Q16, DEMO_PSI16, BFV_T, BFV_GAMMA, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
which = sys.argv[2] if len(sys.argv) > 2 else "both"
n, B = 32768, 64
dev = torch.device("cuda", 0)


def run(qs, psis, label):
    R = len(qs)
    ctx = bfv.BFVContext(n, qs, psis, BFV_T, BFV_GAMMA, device=0)
    g = torch.Generator(device=dev).manual_seed(5)
    qcol = torch.tensor(qs, dtype=torch.int64, device=dev).unsqueeze(1)

    def residues(x):
        return torch.where(x.unsqueeze(0) < 0, qcol + x.unsqueeze(0), x.unsqueeze(0).expand(R, n)).contiguous()

    def ternary():
        return residues(torch.randint(-1, 2, (n,), dtype=torch.int64, device=dev, generator=g))

    def err():
        return residues(torch.round(torch.randn(n, device=dev, generator=g) * 3.2).to(torch.int64))

    sk, e_k = ternary(), err()
    pk = torch.zeros(2, R, n, dtype=torch.int64, device=dev)
    pk[1] = synth(torch, R, n, qs, dev, seed=6)
    ctx.keygen(sk, pk, e_k)
    ub = torch.stack([ternary() for _ in range(B)])
    cb0 = torch.stack([ub, ub]).contiguous()
    eb = torch.stack([torch.stack([err() for _ in range(B)]) for _ in range(2)]).contiguous()
    mb = torch.randint(0, BFV_T, (B, n), dtype=torch.int64, device=dev, generator=g)
    bufs = [cb0.clone() for _ in range(4)]
    torch.cuda.synchronize()
    for i in range(reps):
        ctx.encrypt_batch(bufs[i % 4], pk, eb, mb, B)
    torch.cuda.synchronize()
    for i in range(reps):
        ctx.decrypt_batch(bufs[i % 4], sk, B)
    torch.cuda.synchronize()
    # the complete drivers, one ciphertext: keystream -> samplers -> transforms (demo.cu:275-296 times these)
    sk1 = torch.zeros(R, n, dtype=torch.int64, device=dev)
    pk1 = torch.zeros(2, R, n, dtype=torch.int64, device=dev)
    tmp = torch.zeros(R, n, dtype=torch.int64, device=dev)
    rk = torch.zeros(ctx.keygen_random_bytes, dtype=torch.uint8, device=dev)
    re = torch.zeros(ctx.encrypt_random_bytes, dtype=torch.uint8, device=dev)
    c1 = torch.zeros(2, R, n, dtype=torch.int64, device=dev)
    e1 = torch.zeros(2, R, n, dtype=torch.int64, device=dev)
    m1 = mb[0].contiguous()
    for i in range(reps):
        ctx.keygen_rns(rk, sk1, pk1, tmp, nonce=i)
        ctx.encryption_rns(c1, pk1, re, e1, m1, nonce=i)
        ctx.decrypt(c1, sk1)
    torch.cuda.synchronize()
    print("bfv profile run: %s  n=%d R=%d batch=%d reps=%d" % (label, n, R, B, reps))
    ctx.close()


if which in ("both", "5"):
    run(Q60 + [Q60_SPECIAL], PSI60 + [PSI60_SPECIAL], "4 + 1 primes of 60 bits")
if which in ("both", "16"):
    run(DEMO_Q16, DEMO_PSI16, "16 primes of demo.cu:35-36")
print("done")
