#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_cls2.txt; : > $O
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -q -x -k "cls_linear or block_golden or train_step_small or hip_graph_replay or bit_reproducible or bench_config" 2>&1 | grep -v Warn | tail -6 >> $O
for i in 1 2 3; do
  echo -n "run $i : " >> $O
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])" >> $O
done
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_cls2 -o st --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-side > /dev/null 2>&1
cd $R; grep "cls_linear\|gemm_nt_batched\|gemm_nt_kernel<4" $(find gpurun_out/prof_cls2 -name "*kernel_stats.csv" | head -1) | cut -c1-160 >> $O
rm -rf gpurun_out/prof_cls2
cat $O
