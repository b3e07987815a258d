#!/bin/bash
# PMC passes over the NT GEMM microbench (run on the GPU box)
OUT=$PWD/gpurun_out/pmc_nt; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
cd /tmp
for c in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c -d $OUT/$n -o pmc --output-format csv -- python $R/tools/bench_kernels.py "nt-auto" "nt proj" "nt fc" "nt dfc2" > $OUT/$n.log 2>&1
done
