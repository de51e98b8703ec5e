#!/usr/bin/env python3
"""Throughput of the 30-bit path (old/ntt_30bit.cuh entry points) on its native kernels: n = 4096, 32768, 65536.
Tables and Barrett constants come from the library's own host helpers (no oracle here)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ntt-cuda_amd"))

# (q, psi) per ring size: BFV_Scheme/parameter.h:81-136 (getParams30)
PARAMS30 = {4096: (33538049, 2386), 32768: (19070977, 377), 65536: (13631489, 13)}


def setup30(torch, ntt, n, num, dev, seed=3):
    """-> (a int32 [num][n], q, mu, bit_length, psi table, psi^-1 table) on `dev`"""
    import numpy as np
    q, psi = PARAMS30[n]
    tp, ti = ntt.fillTablePsi128(psi, q, ntt.modinv128(psi, q), n)
    bits = q.bit_length()
    mu = (1 << (2 * bits)) // q
    g = torch.Generator(device=dev).manual_seed(seed)
    a = torch.randint(0, q, (num, n), dtype=torch.int32, device=dev, generator=g)
    tab = torch.from_numpy(tp.astype(np.uint32).view(np.int32)).to(dev)
    tabi = torch.from_numpy(ti.astype(np.uint32).view(np.int32)).to(dev)
    return a, q, mu, bits, tab, tabi


def main():
    import torch
    import ntt_cuda_amd as ntt
    dev = torch.device("cuda", 0)
    for n in (4096, 32768, 65536):
        num = (1 << 27) // n                    # 512 MiB of 32-bit words
        a, q, mu, bits, tab, tabi = setup30(torch, ntt, n, num, dev)
        for _ in range(5):
            ntt.forward30(a, n, q, mu, bits, tab, num)
            ntt.inverse30(a, n, q, mu, bits, tabi, num)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ntt.forward30(a, n, q, mu, bits, tab, num)
        e1.record()
        torch.cuda.synchronize()
        f = e0.elapsed_time(e1) / 20
        e0.record()
        for _ in range(20):
            ntt.inverse30(a, n, q, mu, bits, tabi, num)
        e1.record()
        torch.cuda.synchronize()
        i = e0.elapsed_time(e1) / 20
        gb = num * n * 4 * 2 / 1e9
        print("30-bit n=%d num=%d: forward %.3f ms (%.0f GB/s algorithmic, %.2f M transforms/s)  inverse %.3f ms (%.0f GB/s, %.2f M/s)"
              % (n, num, f, gb / f * 1e3, num / f / 1e3, i, gb / i * 1e3, num / i / 1e3))


if __name__ == "__main__":
    main()
