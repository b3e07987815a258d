#!/bin/bash
# per-kernel times of one configs[1] step in both operand flavours (rocprofv3 --kernel-trace --stats), to see what the fp16 flavour pays
# next to bf16.  Output: gpurun_out/r4_f16_stats_{bf16,f16}.csv
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp; R=$PWD; cd /tmp
for f in bf16 f16; do
  rm -rf $R/gpurun_out/prof_f16cmp_$f
  env PVRL_OPERAND=$f rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_f16cmp_$f -o st --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-side > $R/gpurun_out/r4_f16cmp_$f.log 2>&1
  cp $(find $R/gpurun_out/prof_f16cmp_$f -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r4_f16_stats_$f.csv
  find $R/gpurun_out/prof_f16cmp_$f -name "*.csv" -size +5M -delete
done
cd $R; tail -1 gpurun_out/r4_f16cmp_bf16.log | cut -c1-200; tail -1 gpurun_out/r4_f16cmp_f16.log | cut -c1-200
