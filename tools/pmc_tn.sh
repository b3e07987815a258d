#!/bin/bash
# PMC passes over the TN microbench (run on the GPU box): L2 hit/miss and HBM fetch for the weight-gradient kernels
OUT=$PWD/gpurun_out/pmc_tn; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
cd /tmp
for c in "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES"; do
  n=$(echo $c | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $c -d $OUT/$n -o pmc --output-format csv -- python $R/tools/bench_kernels.py tn > $OUT/$n.log 2>&1
done
