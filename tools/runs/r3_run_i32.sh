#!/bin/bash
python -m pytest tests/test_mvit_gpu.py -m gpu -q -x > gpurun_out/r3_pytest_i32.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_i32.log
V=procedurevrl_amd/csrc/variants/libpvrl_hip_head.so
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
for i in 1 2 3; do
PVRL_LIB_PATH=$V python bench.py $B --arch mvit > gpurun_out/r3_i32_mvit_h_$i.json 2>/dev/null
python bench.py $B --arch mvit > gpurun_out/r3_i32_mvit_n_$i.json 2>/dev/null
done
PVRL_LIB_PATH=$V python tools/probe/mvit_pool_times.py 2>&1 | grep -v amdgpu | tail -n 1
python tools/probe/mvit_pool_times.py 2>&1 | grep -v amdgpu | tail -n 1
tail -n 3 gpurun_out/r3_pytest_i32.log; grep -H -o '"value": [0-9.]*' gpurun_out/r3_i32_*.json
