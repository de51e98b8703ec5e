#!/bin/bash
# round 4, batch A: does overlapping launch boundaries pay?  tools/kbench_r4_base (shipped stagger) and kbench_r4_nostag
# (no start stagger at all) -- one stream against k streams over sub-batches (KB_STREAMS; timed between two events that all
# streams join), two processes each
for p in 1 2; do
  for v in r4_base r4_nostag; do
    echo "== $v process $p: one stream, 1024 polynomials"
    KB_PAIR=1 KB_B2B=4 ./tools/kbench_$v 1024 40 20 30 | grep -E "^pair|^forward|^inverse"
    for cfg in "2 1024" "4 1024" "2 512" "3 768" "4 2048" "8 2048" "2 2048"; do
      set -- $cfg
      KB_STREAMS=$1 ./tools/kbench_$v $2 $((61440 / $2)) 20 30 | grep streams=
    done
  done
done
