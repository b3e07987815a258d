#!/bin/bash
python -m pytest tests/test_mvit_gpu.py -m gpu -q -x > gpurun_out/r3_pytest_ab.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_ab.log
V=procedurevrl_amd/csrc/variants/libpvrl_hip_pattn0.so
PVRL_LIB_PATH=$V python tools/probe/mvit_attn_times.py 2>&1 | grep -v amdgpu > gpurun_out/r3_ab_attn0.txt
python tools/probe/mvit_attn_times.py 2>&1 | grep -v amdgpu > gpurun_out/r3_ab_attn1.txt
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
for i in 1 2 3; do
PVRL_LIB_PATH=$V python bench.py $B --arch mvit > gpurun_out/r3_ab_mvit_x0_$i.json 2>/dev/null
python bench.py $B --arch mvit > gpurun_out/r3_ab_mvit_x1_$i.json 2>/dev/null
done
tail -n 3 gpurun_out/r3_pytest_ab.log; grep -H -o '"value": [0-9.]*' gpurun_out/r3_ab_*.json; paste gpurun_out/r3_ab_attn0.txt gpurun_out/r3_ab_attn1.txt | cut -c1-220
