#!/bin/bash
python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -q > gpurun_out/r3_pytest_h.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_h.log
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
V=procedurevrl_amd/csrc/variants
for i in 1 2 3; do
  PVRL_LIB_PATH=$V/libpvrl_hip_nosfix.so python bench.py $B > gpurun_out/r3_h_nosfix_$i.json 2>/dev/null
  python bench.py $B > gpurun_out/r3_h_sfix_$i.json 2>/dev/null
done
PVRL_LIB_PATH=$V/libpvrl_hip_nosfix.so python bench.py $B --frames 32 --batch 8 > gpurun_out/r3_h_nosfix_t32.json 2>/dev/null
python bench.py $B --frames 32 --batch 8 > gpurun_out/r3_h_sfix_t32.json 2>/dev/null
python tools/bench_kernels.py attn > gpurun_out/r3_h_attn_kernels.txt 2>&1
PVRL_LIB_PATH=$V/libpvrl_hip_nosfix.so python tools/bench_kernels.py attn > gpurun_out/r3_h_attn_kernels_nosfix.txt 2>&1
tail -3 gpurun_out/r3_pytest_h.log; grep -h -o '"value": [0-9.]*' gpurun_out/r3_h_*.json; tail -8 gpurun_out/r3_h_attn_kernels.txt; tail -8 gpurun_out/r3_h_attn_kernels_nosfix.txt
