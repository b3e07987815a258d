#!/bin/bash
# GPU call B: full suite on the buffer-op epilogue, then the NT cache-policy sweep + M sweep + PMC passes
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r3_pytest_b.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_b.log
python tools/probe/nt_cache_policy.py run > gpurun_out/r3_cpol_sweep.txt 2>&1
python tools/probe/nt_cache_policy.py run >> gpurun_out/r3_cpol_sweep.txt 2>&1
python tools/probe/nt_cache_policy.py msweep st0_ld0_gm2 > gpurun_out/r3_msweep.txt 2>&1
tools/pmc_cpol.sh st0_ld0_gm2 sc1_nt_gm2 sc1_ld0_gm2 nt_nt_gm2 sc1_nt_gm8 > gpurun_out/r3_pmc_cpol.log 2>&1
python tools/pmc_cpol_sum.py gpurun_out/pmc_cpol > gpurun_out/r3_pmc_cpol_sum.txt 2>&1
find gpurun_out/pmc_cpol -name "*.csv" -size +2M -delete
tail -4 gpurun_out/r3_pytest_b.log; cat gpurun_out/r3_cpol_sweep.txt | tail -16
