#!/usr/bin/env python3
"""bench.py -- forward+inverse 60-bit NTT throughput at n = 2^15 on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: forward_batch then inverse_batch over `--batch`
polynomials per GPU (default 1024 = BASELINE configs[3]'s per-GPU shard: n=32768, 4 x 60-bit RNS primes;
256 MiB per GPU, i.e. past the Infinity Cache).  Inputs are synthetic uniform residues already resident
in HBM when the timed region starts.  Polynomials are independent, so ranks shard the batch with no
data-path collective (weak scaling); the only collectives are the barrier and the max-over-ranks of the
elapsed time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "ntt-cuda_amd"), os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

# BASELINE configs 2-4: the four largest 60-bit primes = 1 mod 2^16, minimal 2n-th roots
Q60 = [1152921504606584833, 1152921504598720513, 1152921504597016577, 1152921504595968001]
PSI60 = [4443670208963, 100545759574150, 31693996050849, 88651361085495]
# BASELINE configs[4] (BFV at n = 2^15, 4 x 60-bit RNS): the special prime that encryption drops = the next 60-bit prime
# = 1 mod 2^16 below Q60[3], with its minimal primitive 2n-th root; t and gamma as demo.cu:28,93
Q60_SPECIAL, PSI60_SPECIAL = 1152921504595640321, 9679305630873
# a Barrett-INEXACT 60-bit prime = 1 mod 2^17 with a primitive 2^16-th root (tests/params.py INEXACT_PRIMES[60]): the raw_literal leg
Q60_INEXACT, PSI60_INEXACT = 1137833256315125761, 448230823712243253
BFV_T, BFV_GAMMA = 1024, 2305843009213683713
# the reference's published BFV configuration (Article.pdf p26 Table 7: n = 32768, log q = 880, r = 16): demo.cu:35-36
DEMO_Q16 = [18014398506729473, 36028797017456641, 36028797014704129, 36028797014573057, 36028797014376449, 36028797013327873,
            36028797013000193, 36028797012606977, 36028797010444289, 36028797009985537, 36028797005856769, 36028797005529089,
            36028797005135873, 36028797003694081, 36028797003563009, 36028797001138177]
DEMO_PSI16 = [58232959302, 1155186985540, 631260524634, 1526647220035, 455957817523, 1650884166641, 10316746886, 768741990072,
              3911086673862, 5947090524825, 47595902954, 2691682578057, 3903338373, 235185854118, 1769787302793, 3151164484090]
# the reference's getParams tuples (q, psi) used by the CPU-1 / CPU-2 rows of BASELINE.md 3 (parameter.h:38-47,73-77)
REF_4096_58BIT = (288230376135196673, 60193018759093)
REF_4096_25BIT = (33538049, 2386)
REF_32768_55BIT = (36028797017456641, 1155186985540)
HBM_PEAK = 8.0e12            # B/s, MI355X spec (MI355X_MICROARCH.md)
BYTES_PER_TRANSFORM = 2 * 32768 * 8   # one in-place transform reads and writes the polynomial once


def newest_profile(pattern):
    """newest committed profile of a kind: `pattern` holds one %d for the round number (profiles/traffic_r%02d.json ...)"""
    import glob
    import re
    rx = re.compile("^" + re.escape(pattern).replace(re.escape("%02d"), r"(\d+)").replace(re.escape("%d"), r"(\d+)") + "$")
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "*")):
        m = rx.match(os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    return best[1] if best else None


def rocprof_avg_ms(kernel):
    """average launch duration of `kernel` in the newest committed `rocprofv3 --kernel-trace --stats` summary of this workload
    (profiles/rNN_rocprofv3_summary_batch1024.txt, produced by tools/profile.sh 1024): (ms, file) or (None, None)"""
    # preferred: the trace of bench.py itself (`rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-extras`);
    # else the trace of the small driver at the same batch (tools/profile.sh 1024) -- whichever round is newest
    cands = [p_ for p_ in (newest_profile("r%02d_rocprofv3_bench_py_stats.txt"), newest_profile("r%02d_rocprofv3_summary_batch1024.txt")) if p_]
    if not cands:
        return None, None
    import re
    path = max(cands, key=lambda p_: (int(re.match(r"r(\d+)_", os.path.basename(p_)).group(1)), "bench_py" in p_))
    for line in open(path):
        if ("::" + kernel + "<") in line:
            m = re.search(r"avg_ns=([0-9.]+)", line)
            if m:
                return float(m.group(1)) * 1e-6, os.path.relpath(path, ROOT)
    return None, None


PROFILED_BATCH = 1024       # polynomials per launch in the committed rocprofv3 summaries (bench.py's default batch; tools/profile.sh 1024)


class PowerSampler:
    """rocm-smi package power / shader clock sampled in a background thread while a sustained run is in flight (the kernels run
    at the package power cap: profiles/r04_power_cap_and_overlap.txt).  Samples: (seconds since start, sclk MHz, watts)."""

    def __init__(self, period=0.25):
        import threading
        self.period, self.samples, self.stop_flag, self.cap = period, [], False, None
        self.t0 = time.perf_counter()
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _query(self, args):
        return subprocess.run(["rocm-smi"] + args + ["--csv"], capture_output=True, text=True, timeout=10).stdout

    def _run(self):
        import re
        while not self.stop_flag:
            try:
                row = [l for l in self._query(["--showpower", "--showclocks"]).splitlines() if l.startswith("card")]
                if row:
                    f = row[0].split(",")
                    mhz = [int(x) for x in re.findall(r"\((\d+)Mhz\)", row[0])]
                    self.samples.append((time.perf_counter() - self.t0, mhz[2] if len(mhz) > 2 else None, float(f[-1])))
            except Exception:
                pass
            time.sleep(self.period)

    def __enter__(self):
        try:
            import re
            m = re.search(r"card\d+,([0-9.]+)", self._query(["--showmaxpower"]))
            self.cap = float(m.group(1)) if m else None
        except Exception:
            self.cap = None
        self.t0 = time.perf_counter()
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop_flag = True
        self.thread.join(timeout=15)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=1024, help="polynomials per GPU")
    ap.add_argument("--n", type=int, default=32768)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--prewarm-seconds", type=float, default=1.5,
                    help="untimed, time-based pre-warm: back-to-back steps for at least this long before the W warm-up steps (the SMU "
                         "takes about a second of load to settle the clocks under the package-power limit)")
    ap.add_argument("--blocking-sync", action="store_true",
                    help="A/B switch: do not poll an event in front of the bracket's barrier + synchronize (the host then sleeps in "
                         "hipDeviceSynchronize; its wake-up latency lands in the timed region)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo lets two ranks share one GPU for a dry run)")
    ap.add_argument("--end-to-end", action="store_true",
                    help="also time scatter -> forward+inverse -> gather from a rank-0-resident batch (SURVEY.md 8(e), report 2)")
    return ap.parse_args()


def synth_recipe(torch, ctx, num, n, device, seed_base):
    """polynomial y = splitmix64(seed_base + y) mod q[y % P], generated on the device
    (mi355ntt_synth_splitmix; the CPU oracle's synth_batch is the same recipe and the tests compare the two)"""
    a = torch.empty((num, n), dtype=torch.int64, device=device)
    ctx.synth_splitmix(a, num, seed_base)
    return a


def synth(torch, num, n, qs, device, seed):
    """uniform residues for the extras (keys, second operands): 60-bit randoms, one conditional subtraction"""
    g = torch.Generator(device=device).manual_seed(seed)
    a = torch.randint(0, 1 << 60, (num, n), dtype=torch.int64, device=device, generator=g)
    qcol = torch.tensor(qs, dtype=torch.int64, device=device)[torch.arange(num, device=device) % len(qs)].unsqueeze(1)
    return torch.where(a >= qcol, a - qcol, a).contiguous()


def cpu_baseline(n, qs, psis):
    """The oracle (literal restatement of the reference's Barrett CT/GS kernels; the reference ships no CPU NTT)
    timed on this host's cores over a bounded sample of the same workload."""
    import numpy as np
    import oracle_py as oracle
    cores = os.cpu_count() or 1
    so = None
    try:        # host-tuned build of the same source for the timing (the in-tree .so is portable x86-64)
        tmp = tempfile.mkdtemp(prefix="orc_native_")
        so = os.path.join(tmp, "liboracle_native.so")
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-shared", "-std=gnu11", "-o", so,
                               os.path.join(ROOT, "oracle", "ntt_oracle.c"), "-lm"], stderr=subprocess.DEVNULL)
        import ctypes
        lib = ctypes.CDLL(so)
    except Exception:
        lib = oracle.lib()
    prm = oracle.Params(n, qs, psis)
    import ctypes
    u64p, u32p = oracle.u64p, oracle.u32p

    def run(num, threads, min_wall):
        a = oracle.synth_batch(n, num, qs, 1)
        work = a.copy()

        def call(name, tabs):
            f = getattr(lib, name)
            f.restype = None
            f.argtypes = [u64p, ctypes.c_uint, u64p, ctypes.c_uint, ctypes.c_uint, u64p, u64p, u32p, ctypes.c_int]
            f(work.ctypes.data_as(u64p), n, tabs.ctypes.data_as(u64p), num, len(qs), prm.q.ctypes.data_as(u64p),
              prm.mu.ctypes.data_as(u64p), prm.k.ctypes.data_as(u32p), threads)

        call("orc_forward_batch", prm.psi_tabs)        # warm-up + correctness of the round trip
        call("orc_inverse_batch", prm.psiinv_tabs)
        assert np.array_equal(work, a)
        reps, t0 = 0, time.perf_counter()
        while True:
            call("orc_forward_batch", prm.psi_tabs)
            call("orc_inverse_batch", prm.psiinv_tabs)
            reps += 1
            el = time.perf_counter() - t0
            if el >= min_wall:
                break
        return num * reps / el, reps, el

    # one polynomial, one thread, us per forward and per inverse
    def single_poly_us(nn, q, psi, label):
        one = oracle.Params(nn, [q], [psi])
        x = oracle.synth_batch(nn, 1, [q], 1)[0].copy()
        x0 = x.copy()
        f, g = lib.orc_forward, lib.orc_inverse
        for fn in (f, g):
            fn.restype = None
            fn.argtypes = [u64p, ctypes.c_uint, ctypes.c_ulonglong, ctypes.c_ulonglong, ctypes.c_uint, u64p]
        args_f = (x.ctypes.data_as(u64p), nn, int(one.q[0]), int(one.mu[0]), int(one.k[0]), one.psi_tabs[0].ctypes.data_as(u64p))
        args_i = (x.ctypes.data_as(u64p), nn, int(one.q[0]), int(one.mu[0]), int(one.k[0]), one.psiinv_tabs[0].ctypes.data_as(u64p))
        f(*args_f); g(*args_i)
        assert np.array_equal(x, x0)
        reps = max(20, int(2.0e5 // nn) * 4)
        t0 = time.perf_counter()
        for _ in range(reps):
            f(*args_f)
        t1 = time.perf_counter()
        for _ in range(reps):
            g(*args_i)
        t2 = time.perf_counter()
        # (reps forwards then reps inverses of the same buffer: still the identity, checked)
        assert np.array_equal(x, x0)
        return {"config": label, "n": nn, "q": int(q), "bits": int(q).bit_length(), "forward_us": (t1 - t0) / reps * 1e6, "inverse_us": (t2 - t1) / reps * 1e6,
                "pairs_per_s": reps / (t2 - t0)}

    singles = [single_poly_us(4096, *REF_4096_58BIT, "CPU-1 n=4096, 58-bit (parameter.h:43-47)"),
               single_poly_us(4096, *REF_4096_25BIT, "CPU-1 n=4096, 25-bit (parameter.h:38-42, the active set)"),
               single_poly_us(32768, qs[0], psis[0], "CPU-2 n=32768, 60-bit (BASELINE configs[1])"),
               single_poly_us(32768, *REF_32768_55BIT, "CPU-2 n=32768, 55-bit (parameter.h:73-77)")]

    # single thread first (the reference-style scalar rate), then OpenMP over polynomials at a few thread counts;
    # report the best aggregate and the threads it used
    single, _, _ = run(8, 1, 1.0)
    best = (single, 1, 8, 0, 0.0)
    tried = {1: single}
    for th in sorted({max(1, cores // 4), max(1, cores // 2), cores}):
        if th == 1:
            continue
        num = 4 * th - (4 * th) % len(qs)
        rate, reps, el = run(num, th, 2.0)
        tried[th] = rate
        if rate > best[0]:
            best = (rate, th, num, reps, el)
    return {"value": best[0], "unit": "fwd+inv NTT pairs/s", "cores": best[1], "kind": "port",
            "sample": "CPU-3: %d polys (n=%d, %d primes) x %d passes, OpenMP over polynomials, %.1f s wall; host has %d logical CPUs; "
                      "rates by thread count: %s; CPU-1 / CPU-2 (one polynomial, one thread) in single_polynomial_one_thread"
                      % (best[2], n, len(qs), best[3], best[4], cores, ", ".join("%d: %.0f" % (k, v) for k, v in sorted(tried.items()))),
            "single_polynomial_one_thread": singles}


def bfv_round_trip(torch, ntt, n, dev, with_cpu, qs, psis, label):
    """keygen + encrypt + decrypt after the samplers on the GPU, sampled polynomials as inputs;
    the CPU figure is the oracle's literal restatement, one thread (SEAL is not in this image)."""
    from ntt_cuda_amd import bfv
    R = len(qs)
    ctx = bfv.BFVContext(n, qs, psis, BFV_T, BFV_GAMMA, device=dev.index or 0)
    g = torch.Generator(device=dev).manual_seed(5)
    qcol = torch.tensor(qs, dtype=torch.int64, device=dev).unsqueeze(1)

    def residues(x):                       # [n] small signed -> [R][n] residues
        return torch.where(x.unsqueeze(0) < 0, qcol + x.unsqueeze(0), x.unsqueeze(0).expand(R, n)).contiguous()

    def ternary():
        return residues(torch.randint(-1, 2, (n,), dtype=torch.int64, device=dev, generator=g))

    def err():
        return residues(torch.round(torch.randn(n, device=dev, generator=g) * 3.2).to(torch.int64))

    sk0, e_k = ternary(), err()
    pk0 = torch.zeros(2, R, n, dtype=torch.int64, device=dev)
    pk0[1] = synth(torch, R, n, qs, dev, seed=6)
    u, e2 = ternary(), torch.stack([err(), err()])
    c0 = torch.stack([u, u]).contiguous()
    m = torch.randint(0, BFV_T, (n,), dtype=torch.int64, device=dev, generator=g)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, reps=20):
        tot = 0.0
        for i in range(reps + 3):
            args_ = fn(None)
            torch.cuda.synchronize()
            e0.record()
            fn(args_)
            e1.record()
            torch.cuda.synchronize()
            if i >= 3:
                tot += e0.elapsed_time(e1)
        return tot / reps * 1e3, args_

    def keygen(a):
        if a is None:
            return (sk0.clone(), pk0.clone())
        ctx.keygen(a[0], a[1], e_k)

    keygen_us, (sk, pk) = timed(keygen)

    def encrypt(a):
        if a is None:
            return (c0.clone(),)
        ctx.encrypt(a[0], pk, e2, m)

    encrypt_us, (c,) = timed(encrypt)
    c_keep = c.clone()

    def decrypt(a):
        if a is None:
            return (c_keep.clone(),)
        ctx.decrypt(a[0], sk)

    decrypt_us, (cd,) = timed(decrypt)
    off = n * (R - 2)
    assert torch.equal(cd.reshape(-1)[off: off + n], m), "BFV round trip failed"
    out = {"workload": "n=%d, %s, t=%d, 61-bit gamma; drivers after the samplers, one ciphertext" % (n, label, BFV_T),
           "keygen_us": keygen_us, "encrypt_us": encrypt_us, "decrypt_us": decrypt_us, "round_trip_ok": True}

    # like for like with the reference's own timing (demo.cu:275-296 brackets keygen_rns / encryption_rns / decryption_rns, i.e.
    # INCLUDING the Salsa20 keystream and the samplers; Article.pdf p26 Table 7): the complete drivers, one ciphertext per call,
    # GPU time by events over back-to-back calls (fresh nonce each)
    sk1 = torch.zeros(R, n, dtype=torch.int64, device=dev)
    pk1 = torch.zeros(2, R, n, dtype=torch.int64, device=dev)
    tmp = torch.zeros(R, n, dtype=torch.int64, device=dev)
    rk = torch.zeros(ctx.keygen_random_bytes, dtype=torch.uint8, device=dev)
    re_ = torch.zeros(ctx.encrypt_random_bytes, dtype=torch.uint8, device=dev)
    c1 = torch.zeros(2, R, n, dtype=torch.int64, device=dev)
    e1_ = torch.zeros(2, R, n, dtype=torch.int64, device=dev)

    def b2b(fn, reps=50):
        for i in range(5):
            fn(i)
        torch.cuda.synchronize()
        e0.record()
        for i in range(reps):
            fn(100 + i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3

    kg_full = b2b(lambda i: ctx.keygen_rns(rk, sk1, pk1, tmp, nonce=i))
    en_full = b2b(lambda i: ctx.encryption_rns(c1, pk1, re_, e1_, m, nonce=i))
    ctx.encryption_rns(c1, pk1, re_, e1_, m, nonce=7)
    c_full = c1.clone()
    de_full = b2b(lambda i: (c1.copy_(c_full), ctx.decrypt(c1, sk1)))       # (decryption overwrites its input: the copy is inside the bracket, ~2 us)
    ok_full = bool(torch.equal(c1.reshape(-1)[off: off + n], m))
    out["complete_drivers_including_keystream_and_samplers"] = {
        "keygen_rns_us": kg_full, "encryption_rns_us": en_full, "decryption_rns_us": de_full, "round_trip_ok": ok_full,
        "what": "mi355ntt_bfv_keygen_rns / _encryption_rns (Salsa20/20 keystream -> samplers -> transforms) and mi355ntt_bfv_decrypt, one ciphertext per call, "
                "50 calls back to back between two events -- the bracket of demo.cu:275-296"}

    # the batched drivers: 64 ciphertexts per call, layout [2][64][R][n]; same public/secret key, fresh u / e / m each
    B = 64
    ub = torch.stack([ternary() for _ in range(B)])
    cb0 = torch.stack([ub, ub]).contiguous()
    eb = torch.stack([torch.stack([err() for _ in range(B)]) for _ in range(2)]).contiguous()
    mb = torch.randint(0, BFV_T, (B, n), dtype=torch.int64, device=dev, generator=g)

    # throughput figures: K batches back to back between two events, after an untimed pre-warm pass (a region timed right
    # after a host synchronisation reads the clock ramp, not the kernels)
    K = 6

    def stream_rate(fn, bufs):
        f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        warm = [b.clone() for b in bufs[:2]]
        for _ in range(3):
            for w in warm:
                fn(w)
        f0.record()
        for b in bufs:
            fn(b)
        f1.record()
        torch.cuda.synchronize()
        return f0.elapsed_time(f1) * 1e3 / len(bufs)

    enc_bufs = [cb0.clone() for _ in range(K)]
    enc_b_us = stream_rate(lambda c_: ctx.encrypt_batch(c_, pk, eb, mb, B), enc_bufs)
    dec_b_us = stream_rate(lambda c_: ctx.decrypt_batch(c_, sk, B), enc_bufs)
    cbd = enc_bufs[-1]
    assert torch.equal(cbd.reshape(2, B, R, n)[0, :, R - 2], mb), "batched BFV round trip failed"
    out["batch64"] = {"layout": "[2][64][R][n]", "how": "%d batches back to back" % K, "encrypt_us_per_call": enc_b_us, "decrypt_us_per_call": dec_b_us,
                      "encrypt_ciphertexts_per_s": B / (enc_b_us * 1e-6), "decrypt_ciphertexts_per_s": B / (dec_b_us * 1e-6),
                      "encrypt_us_per_ciphertext": enc_b_us / B, "decrypt_us_per_ciphertext": dec_b_us / B, "round_trip_ok": True}
    if with_cpu:
        import numpy as np
        import oracle_py as oracle
        h = lambda x: ntt.to_host(x)
        t0 = time.perf_counter()
        sk_o, pk_o = oracle.bfv_keygen_core(h(sk0), h(pk0), h(e_k), qs, psis, n)
        t1 = time.perf_counter()
        c_o = oracle.bfv_encrypt_core(h(c0), pk_o, h(e2), h(m), qs, psis, n, BFV_T)
        t2 = time.perf_counter()
        m_o = oracle.bfv_decrypt(c_o.reshape(-1), sk_o.reshape(-1)[: (R - 1) * n], qs, psis, n, BFV_T, BFV_GAMMA)
        t3 = time.perf_counter()
        assert np.array_equal(m_o, h(m)) and np.array_equal(c_o.reshape(-1), h(c_keep).reshape(-1))
        out["cpu_oracle_1thread_us"] = {"keygen": (t1 - t0) * 1e6, "encrypt": (t2 - t1) * 1e6, "decrypt": (t3 - t2) * 1e6,
                                        "note": "literal restatement incl. table construction per call; same words as the GPU"}
    ctx.close()
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children BEFORE anything touches the GPU
    (one process per GPU over RCCL, the same command line the driver uses) and relay rank 0's JSON line."""
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver of this pool only supports dmabuf IPC; without it RCCL's (and torch's)
    # cross-process sharing of device memory fails with "hipIpcGetMemHandle: invalid argument" (environment notes of the
    # build image).  The image exports it already; it is pinned here so that the ranks inherit it whatever the caller's
    # shell had, and an explicit setting of the caller's wins.
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    import torch
    import torch.distributed as dist
    import ntt_cuda_amd as ntt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d -- launch with torch.distributed.run --nproc-per-node %d "
                         "(or without a launcher, which spawns the ranks itself)\n" % (args.gpus, world, args.gpus))
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
    # MI355NTT_BENCH_FORCE_PG=1: join a process group even at world size 1, so that the collective bracket of the N > 1 path (RCCL
    # barrier, MAX all-reduce, destroy before the CPU leg) runs on a one-GPU box (tests/test_gpu_round4.py); never set by the driver
    use_pg = world > 1 or os.environ.get("MI355NTT_BENCH_FORCE_PG") == "1"
    if use_pg:
        if world == 1:
            # (MI355NTT_BENCH_FORCE_PG at world size 1: a rendezvous of its own)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + (os.getpid() % 2000)))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        else:
            # world > 1: the launcher's variables or nothing -- a rank that guessed its own port and rank would hang the others
            missing = [v for v in ("MASTER_ADDR", "MASTER_PORT", "RANK", "WORLD_SIZE") if v not in os.environ]
            if missing:
                sys.stderr.write("bench.py: WORLD_SIZE=%d but %s not set -- launch the ranks with torch.distributed.run (or run "
                                 "bench.py --gpus N without a launcher: it spawns them itself)\n" % (world, ", ".join(missing)))
                sys.exit(2)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend=args.backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    n, P, batch = args.n, len(Q60), args.batch
    ctx = ntt.NTTContext(n, Q60, PSI60, device=local)
    # this rank's shard of the global batch: whole polynomials, shard size a multiple of the prime count so
    # polynomial y keeps prime y % P; inputs resident in HBM before timing starts
    a = synth_recipe(torch, ctx, batch, n, dev, seed_base=1 + rank * batch)      # seed = 1 + global polynomial index
    a0 = a.clone()

    def step():
        ctx.forward_batch(a, batch)
        ctx.inverse_batch(a, batch)

    def barrier():
        # the contract's bracket: barrier + torch.cuda.synchronize().  In front of it the host polls an event behind the work queued so
        # far, so that hipDeviceSynchronize() returns without sleeping on an interrupt: one round-6 run read 0.3398 ms per step by wall
        # clock against 0.3316 ms by HIP events over the same 20 steps (0.16 ms of wake-up latency in a 6.8 ms region).  A/B on one box
        # (tools/probe/bench_sync_ab.sh, 4 runs each): wall - events = 2 us per step either way, values within the +-1.5 % a 20-step
        # sample scatters by (the 10-step rounds of one run span 3.03-3.16 M) -- kept because it bounds that latency, not because it gains
        if not args.blocking_sync:
            ev = torch.cuda.Event()
            ev.record()
            while not ev.query():
                pass
        if use_pg:
            dist.barrier()
        torch.cuda.synchronize()

    step()
    torch.cuda.synchronize()
    assert torch.equal(a, a0), "round trip broke the data"
    # cold figure first (reported as an extra): 20 steps straight after the first launch, clocks not yet settled
    t_c = time.perf_counter()
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    cold_pairs_per_s = batch * 20 / (time.perf_counter() - t_c)
    # The SMU needs about a SECOND of load to settle the clocks under the package-power limit (round 5: the driver's 6.8 ms region,
    # 60 ms after the first launch behind a step-counted pre-warm, read 3.00 M pairs/s while the same process sustained 3.12-3.14 M
    # seconds later).  The untimed pre-warm is therefore time-based: back-to-back steps for at least --prewarm-seconds of wall clock,
    # queued in chunks with at most two chunks in flight (the GPU never runs dry, the queue never grows), then the W warm-up steps,
    # flowing straight into the timed region.  What is timed does not change: exactly K steps between barrier + synchronize.
    prewarm_steps, chunk_steps = 0, 100
    pw_ev = []
    t_pw = time.perf_counter()
    while time.perf_counter() - t_pw < args.prewarm_seconds:
        for _ in range(chunk_steps):
            step()
        prewarm_steps += chunk_steps
        pw_ev.append(torch.cuda.Event())
        pw_ev[-1].record()
        if len(pw_ev) >= 3:
            pw_ev[-3].synchronize()
    prewarm_s = time.perf_counter() - t_pw
    for _ in range(args.warmup):
        step()

    # ---- the timed region: EXACTLY K steps between barrier + synchronize on both sides ----
    e_beg, e_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    barrier()
    t0 = time.perf_counter()
    e_beg.record()
    for k in range(args.steps):
        ctx.forward_batch(a, batch)
        ctx.inverse_batch(a, batch)
    e_end.record()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed_min = elapsed
    if use_pg:
        # MAX over ranks of the elapsed time (the contract's clock) and, in the same collective, of its negative: the fastest rank's
        # time -- a straggler GPU shows in the one line as per_rank_pairs_per_s.min well below .max
        tt = torch.tensor([elapsed, -elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, elapsed_min = float(tt[0].item()), -float(tt[1].item())
    step_ms_events = e_beg.elapsed_time(e_end) / args.steps
    assert torch.equal(a, a0)

    # ---- per-kernel launch durations: HIP events on the launch stream (torch's current stream) around K back-to-back
    # launches of each kernel (events between every launch of the step loop would serialise the queue and inflate both)
    def kernel_ms(fn):
        """(ms per launch, shader clock in MHz right behind the same launches: a clock probe -- one wave counting s_memtime
        cycles over 20 us of s_memrealtime -- enqueued behind the closing event)"""
        b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        b0.record()
        for _ in range(args.steps):
            fn()
        b1.record()
        ctx.clock_probe()
        torch.cuda.synchronize()
        return b0.elapsed_time(b1) / args.steps, ctx.probed_clock_mhz()

    fwd_ms, fwd_clock_mhz = kernel_ms(lambda: ctx.forward_batch(a, batch))
    inv_ms, inv_clock_mhz = kernel_ms(lambda: ctx.inverse_batch(a, batch))
    # `a` has now seen K forwards then K inverses: still a valid round trip
    assert torch.equal(a, a0)

    # ---- >= 20 interleaved rounds, median + min.  One round = ROUND_STEPS forward+inverse steps between two
    # events; the rounds are queued back to back (no host synchronisation in between: the clocks stay where they are)
    ROUNDS, ROUND_STEPS = 20, max(1, min(10, args.steps))
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(ROUNDS + 1)]
    for _ in range(10):
        step()
    evs[0].record()
    for r in range(ROUNDS):
        for _ in range(ROUND_STEPS):
            step()
        evs[r + 1].record()
    torch.cuda.synchronize()
    round_ms = sorted(evs[r].elapsed_time(evs[r + 1]) / ROUND_STEPS for r in range(ROUNDS))
    assert torch.equal(a, a0)

    pairs_per_s = world * batch * args.steps / elapsed
    rounds_median = world * batch / (round_ms[ROUNDS // 2] * 1e-3)
    # settled: the contract's region agrees with the median of the 20 rounds timed right behind it within 1.5 % -- the timed
    # region was read at the clocks the chip holds under this load, not on the way there
    settled = abs(pairs_per_s - rounds_median) / rounds_median < 0.015
    dom_name, dom_ms = ("k_forward15", fwd_ms) if fwd_ms >= inv_ms else ("k_inverse15", inv_ms)
    alg_bytes = batch * BYTES_PER_TRANSFORM                       # per launch of either kernel
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9                  # GB/s
    traffic, traffic_source = None, None
    tpath = newest_profile("traffic_r%02d.json")
    if tpath:
        try:
            tj = json.load(open(tpath))
            traffic = tj.get(dom_name, {}).get("hbm_bytes_per_launch")
            traffic_source = "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this workload at commit %s, " \
                             "FETCH_SIZE x 2 per the gfx950 calibration; not collected inside this run -- tests/test_abi_host.py fails when the " \
                             "shipped kernels are no longer the ones that were profiled)" % (os.path.relpath(tpath, ROOT), tj.get("commit", "?"))
        except Exception:
            traffic = None
    # the same fraction from the committed rocprofv3 kernel trace (what a reader can reproduce from profiles/): rocprofv3 reads
    # 5-6 % above the HIP-event time of the same kernel on the same box in every round
    rp_ms, rp_file = rocprof_avg_ms(dom_name)
    # secondary (VALU) ceiling: issue cycles of the kernel's own instruction stream at the measured steady-state cost of
    # each instruction (tools/isa_cost.py over the shipped code object, tools/ubench_issue.hip), one polynomial per CU
    valu = {}
    vpath = newest_profile("valu_ceiling_r%02d.json")
    if vpath:
        try:
            valu = json.load(open(vpath))
        except Exception:
            valu = {}
    cus = torch.cuda.get_device_properties(dev).multi_processor_count

    def ceiling(kname, clock_mhz):
        """transforms/s of kernel `kname` if every issue slot of every SIMD were used: CUs x clock / issue cycles per polynomial
        and CU, at the shader clock the kernel itself sampled during the timed launches"""
        cyc = valu.get(kname, {}).get("cycles_per_polynomial_per_cu")
        return (cus * clock_mhz * 1e6 / cyc) if (cyc and clock_mhz) else None

    ceil_f, ceil_i = ceiling("k_forward15", fwd_clock_mhz), ceiling("k_inverse15", inv_clock_mhz)
    valu_ceiling = ceil_f if dom_name == "k_forward15" else ceil_i
    valu_pairs = (1.0 / (1.0 / ceil_f + 1.0 / ceil_i)) if (ceil_f and ceil_i) else None
    out = {
        "metric": "forward+inverse NTT/s (n=2^15, 60-bit q) per GPU; % HBM roofline",
        "value": pairs_per_s,
        "unit": "fwd+inv NTT pairs/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {"workload": "n=32768, 4x60-bit RNS primes, %d polys/GPU (configs[3] per-GPU shard), forward_batch+inverse_batch, "
                               "inputs resident in HBM" % batch,
                   "n": n, "primes": P, "batch_per_gpu": batch, "global_batch": world * batch, "parallelism": "shard%d" % world},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                     "frac": achieved * 1e9 / HBM_PEAK, "traffic": traffic, "traffic_source": traffic_source, "kernel": dom_name,
                     "avg_launch_ms": dom_ms, "algorithmic_bytes_per_launch": alg_bytes,
                     # the committed profile's own figure: its batch (PROFILED_BATCH) over its average launch time -- not this run's batch
                     "committed_profile": ({"file": rp_file, "kernel": dom_name, "batch": PROFILED_BATCH, "avg_launch_ms": rp_ms,
                                            "frac": PROFILED_BATCH * BYTES_PER_TRANSFORM / (rp_ms * 1e-3) / HBM_PEAK,
                                            "note": "rocprofv3 --kernel-trace --stats of bench.py, another run on another box; `frac` above is this run's HIP-event time"}
                                           if rp_ms else None),
                     "pair_frac_of_hbm_peak": pairs_per_s / world * 2 * BYTES_PER_TRANSFORM / HBM_PEAK,
                     # (repeated here because the driver's parsed copy of the line keeps `roofline` whole: the same step timed as 20 rounds
                     # right behind the contract's region, and whether `value` agrees with their median within 1.5 %)
                     "settled": settled,
                     "rounds_pairs_per_s": {"median": rounds_median, "best": world * batch / (round_ms[0] * 1e-3), "worst": world * batch / (round_ms[-1] * 1e-3)},
                     "valu_ceiling_transforms_per_s": valu_ceiling,
                     "frac_of_valu_ceiling": (batch / (dom_ms * 1e-3) / valu_ceiling) if valu_ceiling else None,
                     "valu_ceiling_pairs_per_s": valu_pairs,
                     "pair_frac_of_valu_ceiling": (pairs_per_s / world / valu_pairs) if valu_pairs else None,
                     "in_kernel_clock_mhz": {"k_forward15": fwd_clock_mhz, "k_inverse15": inv_clock_mhz,
                                             "how": "a clock probe (one wave: s_memtime shader cycles counted over 20 us of the 100 MHz s_memrealtime) enqueued right behind the K back-to-back launches the kernel time is taken from"},
                     "compute_units": cus,
                     "valu_ceiling_source": "%s (tools/valu_ceiling.py: measured issue cycles per instruction summed over the shipped "
                                            "kernels' polynomial loops; a CPU test fails when it drifts from the sources) x compute_units x in_kernel_clock_mhz "
                                            "of this run.  NOT reachable together with the memory traffic: the package sits at its power cap "
                                            "(extras.power_sustained, profiles/r04_power_cap_and_overlap.txt)" % (os.path.relpath(vpath, ROOT) if vpath else None)},
        "kernel_ms": {"k_forward15": fwd_ms, "k_inverse15": inv_ms, "step_by_events": step_ms_events},
        "settled": settled,
        "bracket": "barrier + torch.cuda.synchronize() on both sides" + ("" if args.blocking_sync else
                   "; the host polls an event behind the queued work first, so the synchronize returns without a sleep / interrupt wake-up"),
        "per_rank_pairs_per_s": {"min": batch * args.steps / elapsed, "max": batch * args.steps / elapsed_min,
                                 "what": "slowest and fastest rank over the same K steps (value = world x batch x K / the slowest rank's time)"},
        "prewarm": {"seconds": prewarm_s, "steps": prewarm_steps, "how": "untimed, time-based: chunks of %d steps back to back, at most two chunks queued, until --prewarm-seconds of wall clock have passed; then the W warm-up steps" % chunk_steps},
        "rounds": {"what": "%d rounds of %d forward+inverse steps each, queued back to back, HIP events between rounds (BASELINE.md 2)" % (ROUNDS, ROUND_STEPS),
                   "pairs_per_s_median": rounds_median, "pairs_per_s_best": world * batch / (round_ms[0] * 1e-3),
                   "pairs_per_s_worst": world * batch / (round_ms[-1] * 1e-3),
                   "ms_per_step_median": round_ms[ROUNDS // 2], "ms_per_step_min": round_ms[0], "ms_per_step_max": round_ms[-1],
                   "scope": "this rank" if world > 1 else "the GPU"},
        "cold_20_steps_pairs_per_s": cold_pairs_per_s * world,
    }
    if rank == 0 and world == 1 and not args.no_extras:
        # the reference-signature entry points (forwardNTT_batch / inverseNTT_batch with the caller's tables and moduli,
        # ntt_60bit.cuh:608,652) on the same workload: checked routing (table compared on the device before every call),
        # trusted routing (the caller's promise), and the literal kernels (a hand-made mu keeps the call off the context)
        import numpy as np
        tabs_f = torch.empty((P, n), dtype=torch.int64, device=dev)
        tabs_i = torch.empty((P, n), dtype=torch.int64, device=dev)
        for i in range(P):
            tp, ti = ntt.fillTablePsi128(PSI60[i], Q60[i], ntt.modinv128(PSI60[i], Q60[i]), n)
            tabs_f[i] = torch.from_numpy(tp.view(np.int64))
            tabs_i[i] = torch.from_numpy(ti.view(np.int64))
        mod = ntt.Moduli(Q60)

        def raw_step(m):
            ntt.forwardNTT_batch(a, n, tabs_f, batch, P, m)
            ntt.inverseNTT_batch(a, n, tabs_i, batch, P, m)

        def pairs_rate(fn, reps, prewarm=300):
            # same discipline as the headline: an untimed pre-warm that flows straight into the timed launches -- every host
            # synchronisation (and already a few milliseconds of host work with the queue empty) lets the clocks drop, and they take
            # ~100 steps to come back: tools/raw_trusted_diag.py -- then THREE timed chunks between events, median reported (a long
            # train of unsynchronised launches occasionally stalls once while the runtime recycles its kernel-argument pool)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            for _ in range(prewarm):
                fn()
            ev[0].record()
            for c in range(3):
                for _ in range(reps):
                    fn()
                ev[c + 1].record()
            torch.cuda.synchronize()
            ms = sorted(ev[c].elapsed_time(ev[c + 1]) for c in range(3))
            return batch * reps / (ms[1] * 1e-3)

        raw_checked = pairs_rate(lambda: raw_step(mod), max(40, args.steps))
        assert torch.equal(a, a0)
        ntt.raw_trust_tables(n, tabs_f, mod)
        ntt.raw_trust_tables(n, tabs_i, mod, inverse=True)
        raw_trusted = pairs_rate(lambda: raw_step(mod), max(40, args.steps))
        assert torch.equal(a, a0)
        mu_lit = mod.mu.copy()
        mu_lit[0] -= 1
        lit = ntt.Moduli(Q60, mu=mu_lit, bits=mod.bits)
        scratch = a.clone()

        def lit_step():
            ntt.forwardNTT_batch(scratch, n, tabs_f, batch, P, lit)
            ntt.inverseNTT_batch(scratch, n, tabs_i, batch, P, lit)

        raw_unverifiable = pairs_rate(lit_step, max(4, args.steps // 10), prewarm=20)
        del scratch
        # the reference's own arithmetic at single-pass speed (round 6, kernel class 0): the same call with a Barrett-INEXACT 60-bit
        # modulus (mi355ntt_barrett_is_exact == 0: found by search, tests/params.py INEXACT_PRIMES[60]) -- the context must return the
        # reference's words, q + r included, so every butterfly is singleBarrett written out literally; one read and one write per transform
        raw_literal, lit_valid = None, None
        try:
            assert not ntt.barrett_is_exact(Q60_INEXACT)
            tpi, tii = ntt.fillTablePsi128(PSI60_INEXACT, Q60_INEXACT, ntt.modinv128(PSI60_INEXACT, Q60_INEXACT), n)
            tab_fi = torch.from_numpy(tpi.view(np.int64)).to(dev).reshape(1, n)
            tab_ii = torch.from_numpy(tii.view(np.int64)).to(dev).reshape(1, n)
            modi = ntt.Moduli([Q60_INEXACT])
            bi = synth(torch, batch, n, [Q60_INEXACT], dev, seed=21)

            def inexact_step():
                ntt.forwardNTT_batch(bi, n, tab_fi, batch, 1, modi)
                ntt.inverseNTT_batch(bi, n, tab_ii, batch, 1, modi)

            inexact_step()
            lit_valid = bool(ntt.raw_uses_fast_kernels(n, tab_fi, modi))
            raw_literal = pairs_rate(inexact_step, max(20, args.steps // 2), prewarm=150)
            del bi
        except Exception as exc:            # never let an optional leg break the contract line
            raw_literal = {"error": repr(exc)}
        out["raw_api"] = {"raw_api_pairs_per_s": raw_checked, "raw_api_trusted_pairs_per_s": raw_trusted,
                          "raw_literal_pairs_per_s": raw_literal, "raw_literal_runs_single_pass_kernels": lit_valid,
                          "raw_unverifiable_pairs_per_s": raw_unverifiable,
                          "what": "forwardNTT_batch + inverseNTT_batch through the reference-signature C ABI on the bench workload: "
                                  "checked (per-call device-side table comparison), trusted (mi355ntt_raw_trust_tables); literal = the same "
                                  "checked call on a Barrett-inexact 60-bit modulus (one prime, 1024 polynomials): kernel class 0, the "
                                  "reference's singleBarrett butterflies in the single-pass kernels; unverifiable = a hand-made mu, which "
                                  "keeps the call on the stage kernels (2 passes over memory per transform)"}
        # BASELINE configs[2]: batch 256, pointwise modmul fused (NTT -> (.) -> INTT in one kernel), and configs[1]: batch 1
        b256 = synth(torch, 256, n, Q60, dev, seed=7)
        bh = synth(torch, 256, n, Q60, dev, seed=8)
        ctx.forward_batch(bh, 256)
        # (untimed launches flowing straight into the timed ones, as in the main region: a region timed right after a host
        # synchronisation, or after a few milliseconds of load, reads the clock ramp -- up to 30 % low -- not the kernel)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(400):
            ctx.polymul_batch(b256, bh, 256)
        e0.record()
        for _ in range(20):
            ctx.polymul_batch(b256, bh, 256)
        e1.record()
        torch.cuda.synchronize()
        mul_ms = e0.elapsed_time(e1) / 20
        # the same fused product on the headline batch (1024 polynomials): fraction of the HBM peak on the graded unit
        # (1 310 720 B: forward, product, inverse as three in-place operations) and on what the fused kernel moves (786 432 B)
        bN = synth_recipe(torch, ctx, batch, n, dev, seed_base=5_000_001)
        bhN = synth_recipe(torch, ctx, batch, n, dev, seed_base=6_000_001)
        for _ in range(150):
            ctx.polymul_batch(bN, bhN, batch)
        e0.record()
        for _ in range(40):
            ctx.polymul_batch(bN, bhN, batch)
        e1.record()
        torch.cuda.synchronize()
        mulN_ms = e0.elapsed_time(e1) / 40
        polymul = {"batch": batch, "ms_per_launch": mulN_ms, "products_per_s": batch / (mulN_ms * 1e-3),
                   "polymul_frac": batch * 1310720 / (mulN_ms * 1e-3) / HBM_PEAK,
                   "polymul_frac_fused_bytes": batch * 786432 / (mulN_ms * 1e-3) / HBM_PEAK,
                   "vs_forward_plus_inverse_ms": fwd_ms + inv_ms,
                   "units": "fraction of 8.0 TB/s on 1 310 720 B (SURVEY 8d graded unit) and on 786 432 B (what one fused pass moves)"}
        del bN, bhN

        # ---- the HBM-streaming state: 4096 and 8192 polynomials (configs[3]'s global batch resident on ONE GPU: 2 GiB of 288) ----
        # The headline batch (1024 polynomials = 256 MiB) is the size of the memory-side cache, and the inverse walks the batch
        # downwards to start on what the forward left there; these batches stream from HBM.
        def large_batch(num):
            try:
                big = synth_recipe(torch, ctx, num, n, dev, seed_base=9_000_001)
                bigb = synth_recipe(torch, ctx, num, n, dev, seed_base=19_000_001)
                reps, warm = max(6, 20480 // num), max(12, 61440 // num)

                def rate(fn):
                    for _ in range(warm):
                        fn()
                    e0.record()
                    for _ in range(reps):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    return e0.elapsed_time(e1) / reps

                def pair_():
                    ctx.forward_batch(big, num)
                    ctx.inverse_batch(big, num)

                ref = big[:8].clone()
                p_ms = rate(pair_)
                ok = bool(torch.equal(big[:8], ref))
                f_ms = rate(lambda: ctx.forward_batch(big, num))
                i_ms = rate(lambda: ctx.inverse_batch(big, num))
                m_ms = rate(lambda: ctx.polymul_batch(big, bigb, num))
                del big, bigb
                slow = max(f_ms, i_ms)
                return {"batch": num, "bytes": num * n * 8, "pairs_per_s": num / (p_ms * 1e-3), "pair_ms": p_ms, "forward_ms": f_ms, "inverse_ms": i_ms,
                        "fused_product_ms": m_ms, "fused_products_per_s": num / (m_ms * 1e-3), "round_trip_ok": ok,
                        "ms_per_1024": {"forward": f_ms * 1024 / num, "inverse": i_ms * 1024 / num, "pair": p_ms * 1024 / num, "fused_product": m_ms * 1024 / num},
                        "frac": num * BYTES_PER_TRANSFORM / (slow * 1e-3) / HBM_PEAK,
                        "pair_frac_of_hbm_peak": num / (p_ms * 1e-3) * 2 * BYTES_PER_TRANSFORM / HBM_PEAK,
                        "how": "HIP events around %d back-to-back launches behind %d untimed ones; frac = algorithmic bytes of the slower of the two "
                               "transform kernels / its time / 8 TB/s" % (reps, warm)}
            except Exception as exc:        # never let an optional leg break the contract line
                return {"batch": num, "error": repr(exc)}

        big4096, big8192 = large_batch(4096), large_batch(8192)

        # ---- configs[3] as it would run on an 8-GPU node, DRY RUN on one GPU: the global batch of 8192 polynomials as 8 logical shards
        # of 1024 (one context and three streams per shard, all on this device) through the C ABI's multi-device driver
        # (mi355ntt_shards_*).  Not a scaling number: eight shards share one GPU.  What it shows: the partition, the fork / join and
        # the pipelined scatter -> transform -> gather of a root-resident batch run, and cost nothing against the plain call.
        def logical_shards(world=8, per=1024):
            try:
                total = world * per
                ctxs = [ntt.NTTContext(n, Q60, PSI60, device=local) for _ in range(world)]
                sh = ntt.ShardSet(ctxs, max_polys_per_piece=256)
                full = synth_recipe(torch, ctx, total, n, dev, seed_base=31_000_001)
                ref = full[:: total // 16].clone()
                parts = [full[r * per:(r + 1) * per] for r in range(world)]       # (device-resident shards: views of one allocation)

                def rate(fn, reps=6, warm=6):
                    for _ in range(warm):
                        fn()
                    e0.record()
                    for _ in range(reps):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    return e0.elapsed_time(e1) / reps

                t_res = rate(lambda: sh.transform(ntt.OP_FORWARD_INVERSE, parts, total))
                ok1 = bool(torch.equal(full[:: total // 16], ref))
                t_stg = rate(lambda: sh.scatter_transform_gather(ntt.OP_FORWARD_INVERSE, full, total, chunks=4))
                ok2 = bool(torch.equal(full[:: total // 16], ref))

                def plain():
                    ctx.forward_batch(full, total)
                    ctx.inverse_batch(full, total)

                t_plain = rate(plain)
                sh.close()
                for c_ in ctxs:
                    c_.close()
                del full
                return {"dry_run": "8 logical shards on ONE GPU -- not a scaling number", "global_batch": total, "shards": world,
                        "device_resident_shards_pairs_per_s": total / (t_res * 1e-3), "scatter_transform_gather_pairs_per_s": total / (t_stg * 1e-3),
                        "single_call_pairs_per_s": total / (t_plain * 1e-3), "round_trip_ok": ok1 and ok2}
            except Exception as exc:        # never let an optional leg break the contract line
                return {"error": repr(exc)}

        cfg3 = logical_shards()

        # ---- sustained run with package power / shader clock sampled (the kernels sit at the package power cap) ----
        power = None
        smi_exe = os.path.join(ROOT, "ntt-cuda_amd", "build", "smi_watch")
        smi_proc = None
        try:
            if os.path.exists(smi_exe):       # the SMU's own accounting (rocm_smi gpu_metrics): throttler residencies -> PVIOL / TVIOL
                pr = torch.cuda.get_device_properties(dev)
                bdf = ["%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))] if isinstance(getattr(pr, "pci_bus_id", None), int) else []
                smi_proc = subprocess.Popen([smi_exe, "100", "0"] + bdf, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            with PowerSampler() as ps:
                # ~3 s of headline steps in chunks of 200, at most two chunks queued (the host waits for the event of the chunk before
                # the previous one: the GPU never runs dry, the queue never grows)
                chunks = max(3, int(3.0 / (200 * elapsed / args.steps)))
                cev = [torch.cuda.Event() for _ in range(chunks)]
                t_s = time.perf_counter()
                for c in range(chunks):
                    for _ in range(200):
                        step()
                    cev[c].record()
                    if c >= 2:
                        cev[c - 2].synchronize()
                torch.cuda.synchronize()
                el_s = time.perf_counter() - t_s
                nst = 200 * chunks
            smu = None
            if smi_proc is not None:
                smi_proc.terminate()
                rows = [json.loads(l) for l in smi_proc.communicate(timeout=10)[0].splitlines() if l.startswith("{") and "error" not in l]
                smi_proc = None
                rows = rows[8:] if len(rows) > 12 else rows          # (first 0.8 s dropped, as below)
                if len(rows) >= 2:
                    a_, b_ = rows[0], rows[-1]
                    acc = max(1, b_["acc_counter"] - a_["acc_counter"])
                    smu = {"samples": len(rows), "socket_power_w_mean": sum(r["power_w"] for r in rows) / len(rows),
                           "gfxclk_mhz_mean": sum(r["gfxclk_mhz"] for r in rows) / len(rows), "hotspot_c_max": max(r["hotspot_c"] for r in rows),
                           "pviol_pct": (b_["ppt_acc"] - a_["ppt_acc"]) * 100.0 / acc, "tviol_pct": (b_["thm_acc"] - a_["thm_acc"]) * 100.0 / acc,
                           "hbm_thermal_pct": (b_["hbm_thm_acc"] - a_["hbm_thm_acc"]) * 100.0 / acc, "vr_thermal_pct": (b_["vr_thm_acc"] - a_["vr_thm_acc"]) * 100.0 / acc,
                           "prochot_pct": (b_["prochot_acc"] - a_["prochot_acc"]) * 100.0 / acc,
                           "how": "rocm_smi gpu_metrics every 0.1 s (tools/smi_watch.cpp): PVIOL % = share of the SMU's accumulation cycles in which the "
                                  "package-power-tracking (PPT) limiter held the clock down, TVIOL % the same for the socket thermal limiter"}
            mid = [x for x in ps.samples if 0.8 < x[0] < el_s]
            if mid:
                power = {"pairs_per_s": batch * nst / el_s, "seconds": el_s, "samples": len(mid),
                         "package_power_w_mean": sum(x[2] for x in mid) / len(mid), "package_power_w_max": max(x[2] for x in mid),
                         "sclk_mhz_mean": (sum(x[1] for x in mid if x[1]) / max(1, len([x for x in mid if x[1]]))),
                         "power_cap_w": ps.cap, "smu": smu,
                         "how": "rocm-smi --showpower --showclocks every 0.25 s during %.1f s of back-to-back headline steps (first 0.8 s dropped)" % el_s}
                # the fitted package-power model (tools/power_model.hip -> tools/power_fit.py -> profiles/rNN_power_model_fit.json): what
                # this operating point says about the kernels' own energy per pair, and the pairs/s the cap would admit at other VALU
                # utilisations (the clock the SMU would have to choose falls as the utilisation rises)
                fpath = newest_profile("r%02d_power_model_fit.json")
                if fpath and power["sclk_mhz_mean"] and valu:
                    try:
                        fit = json.load(open(fpath))
                        cyc = valu["k_forward15"]["cycles_per_polynomial_per_cu"] + valu["k_inverse15"]["cycles_per_polynomial_per_cu"]
                        T, f_, w_ = power["pairs_per_s"], power["sclk_mhz_mean"] * 1e-3, power["package_power_w_mean"]
                        S_, A_, al_, M_, be_, f0_ = fit["S_w"], fit["A_w"], fit["alpha"], fit["M_w"], fit["beta"], fit["f0_ghz"]
                        tb = 2 * BYTES_PER_TRANSFORM * 1.03 / 1e12
                        u_ = T * cyc / (cus * f_ * 1e9)
                        e_l = (w_ - S_ - A_ * u_ * (f_ / f0_) ** al_ - M_ * (tb * T) ** be_) / T
                        pred = {}
                        for ut in (0.75, 0.85, 1.0):
                            lo, hi = 1e6, 8e6
                            for _ in range(50):
                                Tm = 0.5 * (lo + hi)
                                fm = min(Tm * cyc / (cus * ut * 1e9), 2.4)
                                um = Tm * cyc / (cus * fm * 1e9)
                                if S_ + A_ * um * (fm / f0_) ** al_ + M_ * (tb * Tm) ** be_ + e_l * Tm > w_:
                                    hi = Tm
                                else:
                                    lo = Tm
                            pred["%.2f" % ut] = lo
                        power["model"] = {"fit": os.path.relpath(fpath, ROOT), "form": fit["model"] + " + E_L x pairs/s",
                                          "valu_utilisation_of_this_run": u_, "E_L_joule_per_pair": e_l,
                                          "pairs_per_s_the_same_power_admits_at_valu_utilisation": pred,
                                          "note": "a 256 MiB batch is partly served by the memory-side cache: its HBM term is an upper bound, E_L absorbs the difference"}
                    except Exception as exc:
                        power["model"] = {"error": repr(exc)}
            else:
                power = {"error": "no rocm-smi samples"}
        except Exception as exc:
            power = {"error": repr(exc)}
        finally:
            if smi_proc is not None:
                smi_proc.kill()
        assert torch.equal(a, a0)
        # the VALU ceiling once more at the clock the chip HOLDS under this load (the clock probe above runs behind the launches, when
        # the load is gone and the clock has already jumped: it reads ~2.4 GHz where rocm-smi shows ~2.25 GHz during the run)
        if isinstance(power, dict) and power.get("sclk_mhz_mean") and valu:
            cf = valu.get("k_forward15", {}).get("cycles_per_polynomial_per_cu")
            ci = valu.get("k_inverse15", {}).get("cycles_per_polynomial_per_cu")
            if cf and ci:
                at = cus * power["sclk_mhz_mean"] * 1e6 / (cf + ci)
                out["roofline"]["valu_ceiling_pairs_per_s_at_sustained_sclk"] = at
                out["roofline"]["pair_frac_of_valu_ceiling_at_sustained_sclk"] = pairs_per_s / at
        one = synth(torch, 1, n, Q60[:1], dev, seed=9)

        def lat(fn):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            e0.record()
            for _ in range(200):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / 200 * 1e3

        pair_us = lat(lambda: (ctx.forward(one, 0), ctx.inverse(one, 0)))
        fwd_us = lat(lambda: ctx.forward(one, 0))
        inv_us = lat(lambda: ctx.inverse(one, 0))
        # the small-batch paths timed from compiled C++ through the C ABI (tools/lat_bench.cpp, built by the package Makefile):
        # per-call time on a stream, as one captured hipGraph, and call + wait; the Python figures beside it
        cpp_lat = None
        lat_exe = os.path.join(ROOT, "ntt-cuda_amd", "build", "lat_bench")
        if os.path.exists(lat_exe):
            try:
                torch.cuda.synchronize()
                r = subprocess.run([lat_exe, "200", "5"], capture_output=True, text=True, timeout=300)
                cpp_lat = json.loads(r.stdout.strip().splitlines()[-1]) if r.returncode == 0 else {"error": (r.stderr or r.stdout)[-300:]}
            except Exception as exc:
                cpp_lat = {"error": repr(exc)}
        # n = 2^16 (the reference's largest ring degree, parameter.h): two half-size transforms per polynomial in ONE launch each
        # (kernels_fast_n16.hip); 512 polynomials = the same 256 MiB as the headline batch
        n16 = None
        try:
            q16 = Q60[0]
            psi16 = next(pw for pw in (pow(x, (q16 - 1) // (2 * 65536), q16) for x in range(2, 1000)) if pow(pw, 65536, q16) == q16 - 1)
            c16 = ntt.NTTContext(65536, [q16], [psi16])
            a16, b16 = synth(torch, 512, 65536, [q16], dev, seed=5), synth(torch, 512, 65536, [q16], dev, seed=6)
            ref16 = a16.clone()

            def rate16(fn):
                for _ in range(60):
                    fn()
                e0.record()
                for _ in range(40):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                return e0.elapsed_time(e1) / 40

            f16 = rate16(lambda: c16.forward_batch(a16, 512))          # (100 forwards in a row: still residues, any words below q do)
            a16.copy_(ref16)
            c16.forward_batch(a16, 512)
            c16.inverse_batch(a16, 512)
            torch.cuda.synchronize()
            ok16 = bool(torch.equal(a16, ref16))
            i16 = rate16(lambda: c16.inverse_batch(a16, 512))
            c16.forward_batch(b16, 512)
            m16 = rate16(lambda: c16.polymul_batch(a16, b16, 512))
            bytes16 = 512 * 65536 * 16
            n16 = {"batch": 512, "forward_ms": f16, "inverse_ms": i16, "fused_product_ms": m16, "round_trip_ok": ok16,
                   "forward_algorithmic_GBps": bytes16 / (f16 * 1e-3) / 1e9, "inverse_algorithmic_GBps": bytes16 / (i16 * 1e-3) / 1e9,
                   "how": "forward: two cooperating workgroups per polynomial (k_forward15_pair); inverse / product: one workgroup per polynomial, coupling stage fused (k_inverse15_split)"}
            c16.close()
            del a16, b16, ref16
        except Exception as exc:            # never let an optional leg break the contract line
            n16 = {"error": repr(exc)}
        out["extras"] = {"config2_fused_polymul_batch256_per_s": 256 / (mul_ms * 1e-3), "config2_fused_polymul_ms": mul_ms,
                         "n65536_batch512": n16,
                         "n32768_batch4096": big4096, "n32768_batch8192": big8192,
                         "configs3_global_batch_as_8_logical_shards": cfg3,
                         "power_sustained": power,
                         "fused_polymul_headline_batch": polymul,
                         "latency_compiled_cpp": cpp_lat,
                         "config1_batch1_fwd_inv_pair_us": pair_us, "config1_batch1_forward_us": fwd_us,
                         "config1_batch1_inverse_us": inv_us,
                         "reference_published_v100_us": {"forward": 39, "inverse": 23, "source": "Article.pdf p25 Table 6 (55-bit q)"},
                         "config4_bfv": bfv_round_trip(torch, ntt, n, dev, not args.no_cpu_baseline, Q60 + [Q60_SPECIAL],
                                                       PSI60 + [PSI60_SPECIAL], "4 x 60-bit RNS + special prime"),
                         "bfv_reference_demo_16_primes": dict(
                             bfv_round_trip(torch, ntt, n, dev, False, DEMO_Q16, DEMO_PSI16, "the 16 primes of demo.cu:35-36 (log q = 880)"),
                             reference_published_v100_us={"keygen": 427.81, "encrypt": 514.73, "decrypt": 246.48, "includes": "samplers",
                                                          "source": "Article.pdf p26 Table 7"})}
    if use_pg:
        out["collective"] = {"backend": dist.get_backend(), "world_size_observed": dist.get_world_size(),
                             "role": "barrier + MAX all-reduce of the timing bracket; scatter / gather of a root-resident batch (end_to_end); no collective on the data path of `value`"}
    # N > 1: from here on the ranks talk to each other outside the timed region (end-to-end leg, teardown).  None of it has ever run
    # across real devices on the build boxes (one GPU each), and the contract line must not depend on it: if the tail has not finished
    # after MI355NTT_BENCH_TAIL_TIMEOUT seconds (default 180) rank 0 prints the line it has -- `value` is complete at this point --
    # with the reason in `end_to_end`, and every rank leaves.
    tail_timer = None
    tail_lock, tail_state = None, {"printed": False}
    if world > 1:
        import threading
        tail_lock = threading.Lock()
        # the contract line as it stands NOW (`value` is complete), serialised before the timer is armed: the timer thread never walks
        # the live dictionary the main thread is still writing to, and exactly one of the two threads prints
        snapshot = dict(out, cpu_baseline=None,
                        end_to_end={"error": "the tail of the run (end-to-end leg / process-group teardown) did not finish in time; the line was printed without it"},
                        tail_timeout=True)
        snapshot_line = json.dumps(snapshot)

        def _bail():
            with tail_lock:
                if tail_state["printed"]:
                    return
                tail_state["printed"] = True
                if rank == 0:
                    print(snapshot_line, flush=True)
            # every rank leaves with 0: rank 0's line is a valid contract line (`value` is complete) and says so itself -- "tail_timeout": true
            # and the reason in `end_to_end` -- so the launcher (and the driver behind it) keeps the measurement instead of a failed run
            os._exit(0)

        tail_timer = threading.Timer(float(os.environ.get("MI355NTT_BENCH_TAIL_TIMEOUT", "180")), _bail)
        tail_timer.daemon = True
        tail_timer.start()
    if args.end_to_end or world > 1:          # N > 1: both figures of SURVEY.md 8(e) in the one line -- `value` (device-resident shards) and end_to_end
        # SURVEY.md 8(e) report 2: the batch lives on rank 0; chunked scatter / transform / gather (ntt_cuda_amd/shard.py)
        try:
            from ntt_cuda_amd import shard
            total = world * batch
            full = synth(torch, total, n, Q60, dev, seed=77) if rank == 0 else None
            reps = max(1, args.steps // 20)

            def tf(piece, count):
                ctx.forward_batch(piece, count)
                ctx.inverse_batch(piece, count)

            if world == 1 and not use_pg:
                e2e = lambda: tf(full.clone(), total)
            else:
                e2e = lambda: shard.scatter_transform_gather(full, total, n, P, tf, chunks=4, src=0, device=dev, inplace=True)
            e2e()
            barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                e2e()
            barrier()
            el = time.perf_counter() - t0
            if use_pg:
                tt = torch.tensor([el], dtype=torch.float64, device=dev)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                el = float(tt.item())
            out["end_to_end"] = {"pairs_per_s": total * reps / el, "ms_per_batch": el / reps * 1e3, "global_batch": total,
                                 "what": "rank-0-resident batch: chunked scatter, forward+inverse per shard, gather back to rank 0"}
        except Exception as exc:            # never let the optional leg break the contract line
            out["end_to_end"] = {"error": repr(exc)}
    ctx.close()
    if use_pg:
        # every rank leaves the collective layer BEFORE rank 0's multi-second CPU leg: nobody sits in an RCCL barrier meanwhile
        dist.barrier()
        dist.destroy_process_group()
    if tail_timer is not None:
        tail_timer.cancel()
        with tail_lock:
            if tail_state["printed"]:        # the timer fired between the last collective and the cancel: its line stands
                return
            tail_state["printed"] = True
    if rank == 0:
        # rank 0's host cores, after the timed region and after the process group is gone; at N = 1 only (the contract asks for it there:
        # at N > 1 the other ranks' processes would share the cores with it, and the scaling runs need not pay 20 s of CPU work each)
        out["cpu_baseline"] = None if (args.no_cpu_baseline or world > 1) else cpu_baseline(n, Q60, PSI60)
        print(json.dumps(out))


if __name__ == "__main__":
    main()
