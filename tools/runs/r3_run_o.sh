#!/bin/bash
python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py tests/test_train_loop_gpu.py -m gpu -q > gpurun_out/r3_pytest_o.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_o.log
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
V=procedurevrl_amd/csrc/variants
for i in 1 2 3; do
  PVRL_LIB_PATH=$V/libpvrl_hip_guarded.so python bench.py $B > gpurun_out/r3_o_guarded_$i.json 2>/dev/null
  python bench.py $B > gpurun_out/r3_o_free_$i.json 2>/dev/null
done
PVRL_LIB_PATH=$V/libpvrl_hip_guarded.so python bench.py $B --frames 32 --batch 8 > gpurun_out/r3_o_guarded_t32.json 2>/dev/null
python bench.py $B --frames 32 --batch 8 > gpurun_out/r3_o_free_t32.json 2>/dev/null
python tools/bench_kernels.py attn > gpurun_out/r3_o_attn.txt 2>&1
PVRL_LIB_PATH=$V/libpvrl_hip_guarded.so python tools/bench_kernels.py attn > gpurun_out/r3_o_attn_guarded.txt 2>&1
tail -3 gpurun_out/r3_pytest_o.log; grep -h -o '"value": [0-9.]*' gpurun_out/r3_o_*.json; tail -n 5 gpurun_out/r3_o_attn.txt gpurun_out/r3_o_attn_guarded.txt
