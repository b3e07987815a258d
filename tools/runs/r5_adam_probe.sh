#!/bin/bash
# AdamW pass geometry probe + the loss-head / weight-cast kernel checks after their rewrites + two bench runs
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_adam_probe.txt; : > $O
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/probe/adam_rate.hip -o /tmp/adam_rate && timeout 300 /tmp/adam_rate >> $O 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "loss or cast_weights or optim" 2>&1 | grep "passed\|failed\|Error" | tail -5 >> $O
for i in 1 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> $O
done
cat $O
