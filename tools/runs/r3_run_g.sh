#!/bin/bash
python -m pytest tests/test_kernels_gpu.py -m gpu -q -x > gpurun_out/r3_pytest_g.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_g.log
python tools/probe/nt_cache_policy.py run > gpurun_out/r3_tails_sweep_g.txt 2>&1
python tools/probe/nt_cache_policy.py msweep tails1 > gpurun_out/r3_msweep_tails1_g.txt 2>&1
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
for i in 1 2 3; do
  PVRL_NT_TAILS=0 python bench.py $B > gpurun_out/r3_g_tails0_$i.json 2>/dev/null
  PVRL_NT_TAILS=1 python bench.py $B > gpurun_out/r3_g_tails1_$i.json 2>/dev/null
done
tail -3 gpurun_out/r3_pytest_g.log; cat gpurun_out/r3_tails_sweep_g.txt; grep "M= 50208" gpurun_out/r3_msweep_tails1_g.txt; grep -h -o '"value": [0-9.]*' gpurun_out/r3_g_tails*.json
