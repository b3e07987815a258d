#!/bin/bash
# last block: projection / MLP on the cls rows only (PVRL_PRUNE_LAST): parity checks, then the step with it on / off
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_prune.txt; : > $O
timeout 2400 python -m pytest tests/test_e2e_gpu.py tests/test_two_rank_gloo_gpu.py -m gpu -q -x 2>&1 | grep "passed\|failed\|Error\|BAD\|assert" | tail -12 >> $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 >> $O
for i in 1 2 3; do for m in 1 0; do
  echo -n "PVRL_PRUNE_LAST=$m : " >> $O
  PVRL_PRUNE_LAST=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('loss'))" >> $O
done; done
cat $O
