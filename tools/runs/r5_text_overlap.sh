#!/bin/bash
# full pre-training step: where the frozen text tower's side stream starts (PVRL_TEXT_OVERLAP = head | encoder | 0); head-engine check first
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_text_overlap.txt; : > $O
timeout 1500 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x -k "pretrain_head_engine or bit_reproducible or e2e_golden" 2>&1 | grep "passed\|failed\|Error" | tail -5 >> $O
for i in 1 2 3; do for m in head encoder 0; do
  echo -n "PVRL_TEXT_OVERLAP=$m : " >> $O
  PVRL_TEXT_OVERLAP=$m timeout 600 python tools/bench_full_step.py --steps 12 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d.get('value'), d.get('ms_per_step'))" >> $O
done; done
cat $O
