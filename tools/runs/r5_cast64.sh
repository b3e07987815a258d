#!/bin/bash
# 64 x 64-tile weight cast: kernel check, then kernel time from a short rocprofv3 run, then the bench
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_cast64.txt; : > $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "cast_weights or input_pipeline" 2>&1 | grep "passed\|failed\|Error" | tail -5 >> $O
export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/cast64_prof -o stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-side > gpurun_out/cast64_prof.log 2>&1
grep -h "cast_weight" gpurun_out/cast64_prof/*kernel_stats.csv | cut -c1-200 >> $O
for i in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> $O
done
cat $O
