#!/bin/bash
# round 4, batch F: energy against issue cycles.  kbench_r4_n1 replaces ONE of the seven multiply-adds of every butterfly (h0 * n1, whose
# second factor is 0xF0000000 for 60-bit near-2^k primes) by a shift and a subtract: +2.8 % VALU issue cycles, about -3 % butterfly energy by
# the single-instruction figures of batch E.  Which one does the kernel follow?  Four processes each, 1024 polynomials (timing only: the lab
# build's twiddle tables are not in the device layout, results are not checked here).
for p in 1 2 3 4; do for v in r4_base r4_n1; do echo "== $v (process $p)"; KB_PAIR=1 KB_B2B=4 ./tools/kbench_$v 1024 40 20 30 | grep -E "^pair|^forward|^inverse"; done; done
