#!/bin/bash
# PMC passes (HBM-side fetch / write bytes, L2 hit rate, MFMA-busy) over NT cache-policy variants (run on the GPU box).
# usage: tools/pmc_cpol.sh <tag> [<tag> ...]    -> gpurun_out/pmc_cpol/<tag>/<counter>/...csv ; summarise with tools/pmc_cpol_sum.py
OUT=$PWD/gpurun_out/pmc_cpol; mkdir -p $OUT; export TMPDIR=/tmp; R=$PWD
cd /tmp
for tag in "$@"; do
  mkdir -p $OUT/$tag
  for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    n=$(echo $c | tr ' ' '_')
    rocprofv3 --kernel-trace --pmc $c -d $OUT/$tag/$n -o pmc --output-format csv -- python $R/tools/probe/nt_cache_policy.py run $tag > $OUT/$tag/$n.log 2>&1
  done
done
find $OUT -name "*kernel_trace.csv" | head -3
du -sh $OUT
