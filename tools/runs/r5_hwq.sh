#!/bin/bash
# round 5: step time vs ROCm's hardware-queue count (GPU_MAX_HW_QUEUES; default 4) with the weight gradients on the main stream (default) / a side stream
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_hwq2.txt; : > $O
for rep in 1 2; do for q in 4 2 1 6; do for env in "PVRL_WGRAD_OVERLAP=0" "PVRL_WGRAD_OVERLAP=0 PVRL_PREFETCH_FUSED=0" "PVRL_WGRAD_OVERLAP=1"; do
  echo -n "GPU_MAX_HW_QUEUES=$q $env : " >> $O
  env GPU_MAX_HW_QUEUES=$q $env timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing > /tmp/b.out 2> /tmp/b.err
  tail -1 /tmp/b.out | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])
except Exception as e: print('FAILED', open('/tmp/b.err').read()[-600:].replace(chr(10),' | '))" >> $O
done; done; done
cat $O
