#!/bin/bash
# order-shaking soak: the GPU suite with the test files in reverse order, then the C++ thread test twenty times
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
{
python -m pytest $(ls tests/test_*.py | sort -r) -m gpu -x -q -p no:cacheprovider 2>&1 | tail -5
ok=0; bad=0
for i in $(seq 1 20); do
  if tests/cpp/threads_test 8 4 > /tmp/tt.out 2> /tmp/tt.err; then ok=$((ok+1)); else bad=$((bad+1)); tail -3 /tmp/tt.out; tail -20 /tmp/tt.err; fi
done
echo "threads_test: $ok ok, $bad failed"
} 2>&1 | tee gpurun_out/soak_r06.txt
