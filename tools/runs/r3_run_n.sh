#!/bin/bash
python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r3_pytest_n.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_n.log
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
V=procedurevrl_amd/csrc/variants
for i in 1 2 3; do
  PVRL_ZERO_UNUSED=1 PVRL_LIB_PATH=$V/libpvrl_hip_nodot2.so python bench.py $B > gpurun_out/r3_n_old_$i.json 2>/dev/null
  PVRL_ZERO_UNUSED=1 python bench.py $B > gpurun_out/r3_n_dot2_$i.json 2>/dev/null
  python bench.py $B > gpurun_out/r3_n_new_$i.json 2>/dev/null
done
tail -3 gpurun_out/r3_pytest_n.log; grep -h -o '"value": [0-9.]*' gpurun_out/r3_n_*.json
