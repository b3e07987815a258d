#!/bin/bash
# MViTv2-S: the last block's projection / MLP on the cls rows only (PVRL_PRUNE_LAST): parity checks, then clips/s with it on / off
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_mvit_prune.txt; : > $O
timeout 2400 python -m pytest tests/test_mvit_gpu.py -m gpu -q -x 2>&1 | grep "passed\|failed\|Error\|BAD\|assert" | tail -12 >> $O
for i in 1 2; do for m in 1 0; do
  echo -n "PVRL_PRUNE_LAST=$m : " >> $O
  PVRL_PRUNE_LAST=$m timeout 900 python bench.py --arch mvit --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('loss'))" >> $O
done; done
cat $O
