#!/bin/bash
# pre-training head with level buffers / grouped stack weight gradients / one weight-cast launch: checks, full-step bench, launch count
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_head_launches.txt; : > $O
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_mvit_gpu.py -m gpu -q -x -k "pretrain or e2e_golden or tfm or head or full" 2>&1 | grep "passed\|failed\|Error\|BAD\|assert" | tail -12 >> $O
for i in 1 2 3; do
  timeout 600 python tools/bench_full_step.py --steps 12 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d.get('value'), d.get('ms_per_step'))" >> $O
done
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_head_trace -o tr --output-format csv -- python $R/tools/bench_full_step.py --steps 4 --warmup 6 > $R/gpurun_out/prof_head_trace.log 2>&1
cd $R
python tools/timeline.py $(find gpurun_out/prof_head_trace -name "*kernel_trace.csv" | head -1) 2>&1 | head -6 >> $O
find gpurun_out/prof_head_trace -name "*.csv" -size +20M -delete
cat $O
