#!/bin/bash
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "gemm_nt" > gpurun_out/r3_pytest_k.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_k.log
NT_STRESS=1 timeout 600 python tools/probe/nt_cache_policy.py run > gpurun_out/r3_pers_sweep.txt 2>&1
timeout 300 python tools/probe/nt_cache_policy.py msweep pers_all > gpurun_out/r3_msweep_pers.txt 2>&1
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
for i in 1 2; do
  PVRL_NT_PERSIST=0 timeout 300 python bench.py $B > gpurun_out/r3_k_pers0_$i.json 2>/dev/null
  PVRL_NT_PERSIST=0x7f timeout 300 python bench.py $B > gpurun_out/r3_k_persall_$i.json 2>/dev/null
  PVRL_NT_PERSIST=0x07 timeout 300 python bench.py $B > gpurun_out/r3_k_pers07_$i.json 2>/dev/null
done
tail -3 gpurun_out/r3_pytest_k.log; cat gpurun_out/r3_pers_sweep.txt; grep "M= 50208\|M= 65536" gpurun_out/r3_msweep_pers.txt; grep -h -o '"value": [0-9.]*' gpurun_out/r3_k_*.json
