#!/bin/bash
python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -q > gpurun_out/r3_pytest_m.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_m.log
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
V=procedurevrl_amd/csrc/variants
for i in 1 2 3; do
  PVRL_LIB_PATH=$V/libpvrl_hip_ieeediv.so python bench.py $B > gpurun_out/r3_m_ieeediv_$i.json 2>/dev/null
  python bench.py $B > gpurun_out/r3_m_rcp_$i.json 2>/dev/null
done
python tools/bench_kernels.py "nt[256x256] fc1" "nt[256x256] dfc2" > gpurun_out/r3_m_kernels.txt 2>&1
PVRL_LIB_PATH=$V/libpvrl_hip_ieeediv.so python tools/bench_kernels.py "nt[256x256] fc1" "nt[256x256] dfc2" > gpurun_out/r3_m_kernels_ieeediv.txt 2>&1
tail -3 gpurun_out/r3_pytest_m.log; grep -h -o '"value": [0-9.]*' gpurun_out/r3_m_*.json; tail -4 gpurun_out/r3_m_kernels.txt gpurun_out/r3_m_kernels_ieeediv.txt
