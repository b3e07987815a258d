#!/bin/bash
# fp32 small GEMMs with K % 128 == 0 on the fp32 MFMA kernel of cls_chain.hip: checks, then the step and the full step with it on / off
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_f32_small.txt; : > $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "f32_small or cls_linear or loss" 2>&1 | grep "passed\|failed\|Error\|BAD" | tail -5 >> $O
timeout 1500 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x -k "pretrain_head_engine or e2e_golden or bench_config or train_step_small" 2>&1 | grep "passed\|failed\|Error\|BAD" | tail -5 >> $O
for i in 1 2 3; do for m in 1 0; do
  echo -n "PVRL_F32_SMALL_MFMA=$m : step " >> $O
  PVRL_F32_SMALL_MFMA=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], end='   full step ')" >> $O
  PVRL_F32_SMALL_MFMA=$m timeout 600 python tools/bench_full_step.py --steps 12 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d.get('value'), d.get('ms_per_step'))" >> $O
done; done
cat $O
