#!/bin/bash
python -m pytest tests/test_mvit_gpu.py -m gpu -q -x > gpurun_out/r3_pytest_dg.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_dg.log
V=procedurevrl_amd/csrc/variants/libpvrl_hip_dgs0.so
PVRL_LIB_PATH=$V python tools/probe/mvit_pool_times.py 2>&1 | grep -v amdgpu > gpurun_out/r3_dg_pool0.txt
python tools/probe/mvit_pool_times.py 2>&1 | grep -v amdgpu > gpurun_out/r3_dg_pool1.txt
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
for i in 1 2 3; do
PVRL_LIB_PATH=$V python bench.py $B --arch mvit > gpurun_out/r3_dg_mvit_s0_$i.json 2>/dev/null
python bench.py $B --arch mvit > gpurun_out/r3_dg_mvit_s1_$i.json 2>/dev/null
done
tail -n 3 gpurun_out/r3_pytest_dg.log; grep -H -o '"value": [0-9.]*' gpurun_out/r3_dg_*.json; paste <(cut -c1-62 gpurun_out/r3_dg_pool0.txt) <(cut -c28-62 gpurun_out/r3_dg_pool1.txt) | sed -n '1,5p;15,17p'
