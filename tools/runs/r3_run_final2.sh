#!/bin/bash
rm -rf gpurun_out/prof_r3_mvit
BENCH_ARGS="--arch mvit" tools/profile_round.sh r3_mvit > gpurun_out/r3_prof_mvit.log 2>&1
export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf $R/gpurun_out/prof_r3_mvit_trace
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r3_mvit_trace -o tr --output-format csv -- python $R/bench.py --arch mvit --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-side > $R/gpurun_out/r3_prof_mvit_trace.log 2>&1
cd $R
python tools/timeline.py $(find gpurun_out/prof_r3_mvit_trace -name "*kernel_trace.csv" | head -1) > gpurun_out/r3_timeline_mvit.txt 2>&1
find gpurun_out/prof_r3_mvit_trace -name "*.csv" -size +20M -delete
python tools/probe/mvit_gemm_times.py 2>&1 | grep -v amdgpu > gpurun_out/r3_final_mvit_shapes.txt
tools/runs/r3_run_full.sh
