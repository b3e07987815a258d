#!/bin/bash
# end-of-round profiles: rocprofv3 stats + the three separate PMC passes (tools/profile_round.sh) for configs[1], [3] (T = 32) and
# [4] (MViTv2-S), and the kernel-trace timeline of one step of configs[1].  Summarise with tools/summarize_prof.py / summarize_pmc.py.
# usage (on the GPU box, via gpurun): bash tools/runs/r4_prof.sh [main|all]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
W=${1:-main}
rm -rf gpurun_out/prof_r4 gpurun_out/prof_r4_t32 gpurun_out/prof_r4_mvit gpurun_out/prof_r4_trace
tools/profile_round.sh r4 > gpurun_out/r4_prof_main.log 2>&1
if [ "$W" = all ]; then
  BENCH_ARGS="--arch mvit" tools/profile_round.sh r4_mvit > gpurun_out/r4_prof_mvit.log 2>&1
  BENCH_ARGS="--frames 32 --batch 8" tools/profile_round.sh r4_t32 > gpurun_out/r4_prof_t32.log 2>&1
fi
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r4_trace -o tr --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-side > $R/gpurun_out/r4_prof_trace.log 2>&1
cd $R
python tools/timeline.py $(find gpurun_out/prof_r4_trace -name "*kernel_trace.csv" | head -1) > gpurun_out/r4_timeline.txt 2>&1
find gpurun_out/prof_r4_trace -name "*.csv" -size +20M -delete
du -sh gpurun_out/prof_r4*; head -12 gpurun_out/r4_timeline.txt
