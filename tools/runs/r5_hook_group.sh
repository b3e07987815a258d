#!/bin/bash
# round 5: hook groups (data-parallel path): tests of the staged / hooked backward, then the N > 1 machinery on one GPU with PVRL_HOOK_GROUP = 1 / 3 / 4
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_hook_group.txt; : > $O
timeout 1800 python -m pytest tests/test_e2e_gpu.py tests/test_two_rank_gloo_gpu.py tests/test_nccl_single_rank_gpu.py tests/test_launcher_gpu.py tests/test_mvit_gpu.py -m gpu -q -k "hip_graph_replay or two_rank or nccl or bench or mvit" 2>&1 | grep "passed\|failed" | tail -3 >> $O
for i in 1 2; do for g in 3 1 4; do
  echo -n "PVRL_HOOK_GROUP=$g --world1-rccl : " >> $O
  PVRL_HOOK_GROUP=$g timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing --world1-rccl 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> $O
done; done
cat $O
