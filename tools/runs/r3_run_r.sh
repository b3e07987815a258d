#!/bin/bash
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm" > gpurun_out/r3_pytest_r.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_r.log
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
V=procedurevrl_amd/csrc/variants
for i in 1 2 3; do
  PVRL_LIB_PATH=$V/libpvrl_hip_nofill2.so python bench.py $B > gpurun_out/r3_r_nofill_$i.json 2>/dev/null
  python bench.py $B > gpurun_out/r3_r_fill2_$i.json 2>/dev/null
done
for i in 1 2; do
  PVRL_LIB_PATH=$V/libpvrl_hip_nofill2.so python bench.py $B --arch mvit > gpurun_out/r3_r_mvit_nofill_$i.json 2>/dev/null
  python bench.py $B --arch mvit > gpurun_out/r3_r_mvit_fill2_$i.json 2>/dev/null
done
tail -3 gpurun_out/r3_pytest_r.log; grep -H -o '"value": [0-9.]*' gpurun_out/r3_r_*.json
