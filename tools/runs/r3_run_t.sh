#!/bin/bash
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k gemm > gpurun_out/r3_pytest_t.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_t.log
PVRL_NT_WIDE=0 python tools/probe/mvit_gemm_times.py > gpurun_out/r3_t_shapes_wide0.txt 2>&1
PVRL_NT_WIDE=1 python tools/probe/mvit_gemm_times.py > gpurun_out/r3_t_shapes_wide1.txt 2>&1
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
for i in 1 2; do
PVRL_NT_WIDE=0 python bench.py $B --arch mvit > gpurun_out/r3_t_mvit_wide0_$i.json 2>/dev/null
PVRL_NT_WIDE=1 python bench.py $B --arch mvit > gpurun_out/r3_t_mvit_wide1_$i.json 2>/dev/null
done
tail -n 3 gpurun_out/r3_pytest_t.log; grep -H -o '"value": [0-9.]*' gpurun_out/r3_t_*.json
