#!/bin/bash
# the last block's spatial attention on the cls query only (csrc/attn_cls.hip, PVRL_PRUNE_ATTN): kernel + e2e checks, step on / off, kernel times
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_attn_cls.txt; : > $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attn_cls" 2>&1 | grep "passed\|failed\|Error\|BAD\|assert" | tail -8 >> $O
timeout 2400 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x 2>&1 | grep "passed\|failed\|Error\|BAD\|assert" | tail -8 >> $O
for i in 1 2 3; do for m in 1 0; do
  echo -n "PVRL_PRUNE_ATTN=$m : " >> $O
  PVRL_PRUNE_ATTN=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('loss'))" >> $O
done; done
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_attn_cls -o st --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-side > $R/gpurun_out/prof_attn_cls.log 2>&1
cd $R
grep -h "attn_cls" $(find gpurun_out/prof_attn_cls -name "*kernel_stats.csv") | cut -c1-160 >> $O
find gpurun_out/prof_attn_cls -name "*trace.csv" -delete
cat $O
