"""probe: n = 2^16 on a Barrett-inexact 60-bit modulus: us per forward + inverse -- kernel class 0 (two half-size transforms around the literal
coupling stage) against the stage-per-launch kernels, which a context of more than 8 primes still takes at this ring degree"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "ntt-cuda_amd"), os.path.join(ROOT, "tests")]
import ntt_cuda_amd as ntt
import params as P
n = 65536
q, roots = P.INEXACT_PRIMES[60]
for reps_of_q in (1, 9):
  ctx = ntt.NTTContext(n, [q] * reps_of_q, [roots[n]] * reps_of_q)
  for num in ((1, 16, 64, 128, 256, 512) if reps_of_q == 1 else (9, 63, 126, 252, 504)):
      a = torch.randint(0, q, (num, n), dtype=torch.int64, device="cuda")
      def pair():
          ctx.forward_batch(a, num); ctx.inverse_batch(a, num)
      for _ in range(20): pair()
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(50): pair()
      e1.record(); torch.cuda.synchronize()
      print("%d prime(s): %s  n 65536 batch %4d  pair %9.1f us" % (reps_of_q, "stage kernels" if ctx.literal_routing and reps_of_q > 8 else "class 0", num, e0.elapsed_time(e1) / 50 * 1e3), flush=True)
