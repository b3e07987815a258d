#!/bin/bash
# round 5: full pre-training step with the text tower on its side stream, at 1 / 2 / 4 hardware queues (robustness) and overlap off
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_full_q.txt; : > $O
for q in 4 1 2 4; do for ov in 1 0; do
  echo -n "GPU_MAX_HW_QUEUES=$q PVRL_TEXT_OVERLAP=$ov : " >> $O
  GPU_MAX_HW_QUEUES=$q PVRL_TEXT_OVERLAP=$ov timeout 600 python tools/bench_full_step.py --steps 10 --warmup 6 > /tmp/b.out 2> /tmp/b.err
  tail -1 /tmp/b.out | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])
except Exception as e: print('FAILED', open('/tmp/b.err').read()[-400:].replace(chr(10),' | '))" >> $O
done; done
cat $O
