#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_ab_batch.txt; : > $O
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py tests/test_two_rank_gloo_gpu.py -m gpu -q -x -k "small_batched or layernorm or block_golden or train_step_small or hip_graph_replay or bit_reproducible or e2e_golden or droppath or two_rank" 2>&1 | grep "passed\|failed\|Error" | tail -5 >> $O
for i in 1 2 3; do for b in 1 0; do
  echo -n "PVRL_BATCH_FUSED=$b : " >> $O
  PVRL_BATCH_FUSED=$b timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> $O
done; done
cat $O
