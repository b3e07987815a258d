#!/bin/bash
python -m pytest tests/test_mvit_gpu.py -m gpu -q -x > gpurun_out/r3_pytest_rel.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_rel.log
PVRL_RELQ_LDS=0 python tools/probe/mvit_attn_times.py 2>&1 | grep -v amdgpu > gpurun_out/r3_rel_attn0.txt
python tools/probe/mvit_attn_times.py 2>&1 | grep -v amdgpu > gpurun_out/r3_rel_attn1.txt
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
for i in 1 2 3; do
PVRL_RELQ_LDS=0 python bench.py $B --arch mvit > gpurun_out/r3_rel_mvit_l0_$i.json 2>/dev/null
python bench.py $B --arch mvit > gpurun_out/r3_rel_mvit_l1_$i.json 2>/dev/null
done
tail -n 3 gpurun_out/r3_pytest_rel.log; grep -H -o '"value": [0-9.]*' gpurun_out/r3_rel_*.json; paste <(cut -c1-90 gpurun_out/r3_rel_attn0.txt) <(cut -c40-90 gpurun_out/r3_rel_attn1.txt)
