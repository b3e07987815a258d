#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_mvit_dxn.txt; : > $O
timeout 1500 python -m pytest tests/test_mvit_gpu.py -m gpu -q 2>&1 | tail -4 >> $O
for i in 1 2; do for m in 1 0; do
  echo -n "PVRL_MVIT_DXN16=$m : " >> $O
  PVRL_MVIT_DXN16=$m timeout 600 python bench.py --arch mvit --steps 10 --warmup 3 --no-cpu-baseline --no-side --no-kernel-timing --no-parity-probe 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> $O
done; done
cat $O
