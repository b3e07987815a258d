#!/bin/bash
python tools/probe/nt_cache_policy.py run > gpurun_out/r3_cpol_sweep2.txt 2>&1
tools/pmc_cpol.sh st0_ld0_gm2 st0_ld0_gm4 sc1_nt_gm2 > gpurun_out/r3_pmc_cpol.log 2>&1
python tools/pmc_cpol_sum.py gpurun_out/pmc_cpol > gpurun_out/r3_pmc_cpol_sum.txt 2>&1
find gpurun_out/pmc_cpol -name "*.csv" -size +2M -delete
python -m pytest tests/test_mvit_gpu.py -m gpu -q -k timed > gpurun_out/r3_mvit_timed_bf16.log 2>&1
PVRL_OPERAND=f16 python -m pytest tests/test_mvit_gpu.py -m gpu -q -k timed > gpurun_out/r3_mvit_timed_f16.log 2>&1
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
V=procedurevrl_amd/csrc/variants
for i in 1 2; do
  python bench.py $B > gpurun_out/r3_ab_default_$i.json 2>/dev/null
  PVRL_LIB_PATH=$V/libpvrl_hip_gm4.so python bench.py $B > gpurun_out/r3_ab_gm4_$i.json 2>/dev/null
  PVRL_LIB_PATH=$V/libpvrl_hip_gm3.so python bench.py $B > gpurun_out/r3_ab_gm3_$i.json 2>/dev/null
done
BENCH_ARGS="--frames 32 --batch 8" tools/profile_round.sh r3_t32 > gpurun_out/r3_prof_t32.log 2>&1
cat gpurun_out/r3_cpol_sweep2.txt; grep -h -o '"value": [0-9.]*' gpurun_out/r3_ab_*.json
