#!/bin/bash
# Round-3 batch D: what each part of the memory traffic costs in the launch geometry of the persistent kernels (results are WRONG
# in these builds).  c3 = shipped; m_ncx = no butterflies, no workgroup exchange (the pure memory pattern); m_c = no global memory
# at all (compute, LDS, barriers only); m_cns = m_c without the start stagger; m_ng = the 8-byte coalesced accesses removed (forward:
# no loads, row stores stay; inverse: row loads stay, no stores); m_nr = the 16-byte row accesses removed (forward: loads only;
# inverse: stores only).
for p in 1 2; do for v in c3 m_ncx m_c m_cns m_ng m_nr; do echo "== $v (process $p)"; KB_PAIR=1 KB_B2B=4 ./tools/kbench_$v 1024 40 20 30; done; done
