#!/bin/bash
python -m pytest tests/test_kernels_gpu.py tests/test_mvit_gpu.py -m gpu -q > gpurun_out/r3_pytest_v.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_v.log
python tools/probe/mvit_gemm_times.py > gpurun_out/r3_v_shapes.txt 2>&1
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
V=procedurevrl_amd/csrc/variants/libpvrl_hip_tnpad0.so
for i in 1 2 3; do
PVRL_LIB_PATH=$V python bench.py $B --arch mvit > gpurun_out/r3_v_mvit_pad0_$i.json 2>/dev/null
python bench.py $B --arch mvit > gpurun_out/r3_v_mvit_pad1_$i.json 2>/dev/null
done
tail -n 3 gpurun_out/r3_pytest_v.log; grep -H -o '"value": [0-9.]*' gpurun_out/r3_v_*.json; tail -n 1 gpurun_out/r3_v_shapes.txt
