#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_mid_m.txt; : > $O
for rep in 1 2 3; do for m in 2048 4096; do
  echo -n "PVRL_NT_MID_M=$m : " >> $O
  PVRL_NT_MID_M=$m timeout 600 python tools/bench_full_step.py --steps 10 --warmup 6 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> $O
done; done
cat $O
