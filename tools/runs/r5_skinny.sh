#!/bin/bash
# few-row NT GEMM (csrc/gemm_nt_skinny.h): kernel + head checks, then the full pre-training step with and without it
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_skinny.txt; : > $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm_nt" 2>&1 | grep "passed\|failed\|Error\|BAD" | tail -8 >> $O
timeout 1500 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x -k "pretrain_head_engine or e2e_golden or tfm" 2>&1 | grep "passed\|failed\|Error\|BAD" | tail -8 >> $O
for i in 1 2 3; do for m in 1 0; do
  echo -n "PVRL_NT_SKINNY=$m : " >> $O
  PVRL_NT_SKINNY=$m timeout 600 python tools/bench_full_step.py --steps 12 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d.get('value'), d.get('ms_per_step'))" >> $O
done; done
cat $O
