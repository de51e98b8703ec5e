#!/bin/bash
# round 5, batch C: the GPU suite on the round-5 tree (threads, moduli sweep, raw per-prime routing, element-wise wrappers on
# word-aligned pointers), bench.py, and the BFV profile of the shipped library.
O=gpurun_out/r05c
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu"
python3 -m pytest tests -x -q -m gpu --durations=15 2>&1 | tail -40
echo "== bench.py"
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.err; python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05c/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"])
ex = d.get("extras", {})
for k in ("n32768_batch4096", "n32768_batch8192", "power_sustained"):
    print(k, json.dumps(ex.get(k))[:1500])
print("bfv", json.dumps(ex.get("config4_bfv"))[:2500])
print("bfv16", json.dumps(ex.get("bfv_reference_demo_16_primes"))[:2500])
PY
echo "== BFV profile"
for set in 5 16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/bfv$set/trace -- python3 tools/prof_driver_bfv.py 20 $set > $O/bfv$set.log 2>&1
  python3 tools/prof_summary_bfv.py $O/bfv$set $set > $O/bfv${set}_summary.txt 2>&1
  grep -v "at::native\|rocclr" $O/bfv${set}_summary.txt
done
