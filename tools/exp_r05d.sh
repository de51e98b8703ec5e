#!/bin/bash
# round 5, batch D: GPU suite on the tree with the graceful pair watchdog, raw per-prime routing and the class 5 / 3 split kernels
# of n = 2^16; the n = 2^16 class sweep with and without those kernels.
O=gpurun_out/r05d
mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu"
python3 -m pytest tests -q -m gpu --durations=12 2>&1 | tail -45
echo "== n = 65536 classes (round-5 kernels)"
python3 tools/sweep_classes16.py 512
echo "== n = 65536 classes (classes 5 / 3 folded into 4 / 2, as round 4)"
MI355NTT_N16_NO_CLASS35=1 python3 tools/sweep_classes16.py 512
echo "== n = 32768 classes"
python3 tools/sweep_classes.py 1024
