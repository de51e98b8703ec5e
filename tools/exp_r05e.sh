#!/bin/bash
# round 5, batch E: the GPU suite on the final kernels, the class sweeps, bench.py (also under rocprofv3 --kernel-trace --stats),
# the PMC traffic passes, the BFV profile.
O=gpurun_out/r05e
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu"
python3 -m pytest tests -q -m gpu --durations=8 2>&1 | tail -25
echo "== n = 32768 classes"
python3 tools/sweep_classes.py 1024 | tee $O/classes15.txt
echo "== n = 65536 classes"
python3 tools/sweep_classes16.py 512 | tee $O/classes16.txt
echo "== pair forward A/B"
python3 tools/probe/pair16_ab.py; MI355NTT_NO_PAIR16=1 python3 tools/probe/pair16_ab.py
echo "== bench.py"
python3 bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r05e/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], d["kernel_ms"])
ex = d.get("extras", {})
print("4096", ex["n32768_batch4096"]["pairs_per_s"], "8192", ex["n32768_batch8192"]["pairs_per_s"], "n16", ex["n65536_batch512"])
print("power", json.dumps(ex["power_sustained"])[:1200])
PY
echo "== rocprofv3 of bench.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/profbench/trace -- python3 bench.py --no-cpu-baseline --no-extras > $O/profbench_line.json 2> $O/profbench.log
python3 tools/prof_summary.py $O/profbench > $O/profbench_summary.txt 2>&1; grep -E "k_forward15|k_inverse15" $O/profbench_summary.txt | head
echo "== rocprofv3 at 8192 polynomials"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof8192/trace -- python3 tools/prof_driver.py 8192 30 > $O/prof8192.log 2>&1
python3 tools/prof_summary.py $O/prof8192 > $O/prof8192_summary.txt 2>&1; grep -E "k_forward15|k_inverse15|k_polymul15" $O/prof8192_summary.txt | head
echo "== traffic"
mkdir -p gpurun_out/traffic; bash tools/profile_traffic.sh > gpurun_out/traffic/summary.txt 2>&1; tail -20 gpurun_out/traffic/summary.txt
