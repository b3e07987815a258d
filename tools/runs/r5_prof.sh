#!/bin/bash
# end-of-round profiles: rocprofv3 stats + the three separate PMC passes (tools/profile_round.sh) for configs[1] (and, with `all`, configs[3]
# T = 32 and configs[4] MViTv2-S), the kernel-trace timeline of one step of configs[1] and of the full pre-training step.
# usage (on the GPU box, via gpurun): bash tools/runs/r5_prof.sh [main|all]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=${1:-main}
rm -rf gpurun_out/prof_r5 gpurun_out/prof_r5_t32 gpurun_out/prof_r5_mvit gpurun_out/prof_r5_trace gpurun_out/prof_r5_trace_full
tools/profile_round.sh r5 > gpurun_out/r5_prof_main.log 2>&1
if [ "$W" = all ]; then
  BENCH_ARGS="--arch mvit" tools/profile_round.sh r5_mvit > gpurun_out/r5_prof_mvit.log 2>&1
  BENCH_ARGS="--frames 32 --batch 8" tools/profile_round.sh r5_t32 > gpurun_out/r5_prof_t32.log 2>&1
fi
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r5_trace -o tr --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-side > $R/gpurun_out/r5_prof_trace.log 2>&1
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r5_trace_full -o tr --output-format csv -- python $R/tools/bench_full_step.py --steps 4 --warmup 6 > $R/gpurun_out/r5_prof_trace_full.log 2>&1
cd $R
python tools/timeline.py $(find gpurun_out/prof_r5_trace -name "*kernel_trace.csv" | head -1) > gpurun_out/r5_timeline.txt 2>&1
python tools/timeline.py $(find gpurun_out/prof_r5_trace_full -name "*kernel_trace.csv" | head -1) > gpurun_out/r5_timeline_full.txt 2>&1
find gpurun_out/prof_r5_trace gpurun_out/prof_r5_trace_full -name "*.csv" -size +20M -delete
du -sh gpurun_out/prof_r5*; head -12 gpurun_out/r5_timeline.txt; head -12 gpurun_out/r5_timeline_full.txt
