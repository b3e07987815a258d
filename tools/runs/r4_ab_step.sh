#!/bin/bash
# same-box A/B of the training step for one environment switch: three interleaved pairs of
#   <VAR>=0 / <VAR>=1 python bench.py --steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing
# usage: bash tools/runs/r4_ab_step.sh PVRL_ATTN_BWD_FUSED
cd "${GRAFT_REPO_ROOT:-/root/repo}"
V=${1:-PVRL_ATTN_BWD_FUSED}; O=gpurun_out/r4_ab_$V.txt; : > $O
for i in 1 2 3; do
  for x in 0 1; do
    env $V=$x python bench.py --steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V=$x run $i:', d['value'], d['ms_per_step'])" | tee -a $O
  done
done
