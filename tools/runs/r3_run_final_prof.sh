#!/bin/bash
# end-of-round profiles: rocprofv3 stats + PMC passes for configs[1], [3] (T = 32) and [4] (MViTv2-S), kernel-trace timelines of [1] and [4]
rm -rf gpurun_out/prof_r3 gpurun_out/prof_r3_t32 gpurun_out/prof_r3_mvit
tools/profile_round.sh r3 > gpurun_out/r3_prof_main.log 2>&1
BENCH_ARGS="--arch mvit" tools/profile_round.sh r3_mvit > gpurun_out/r3_prof_mvit.log 2>&1
BENCH_ARGS="--frames 32 --batch 8" tools/profile_round.sh r3_t32 > gpurun_out/r3_prof_t32.log 2>&1
export TMPDIR=/tmp; R=$PWD; cd /tmp
rm -rf $R/gpurun_out/prof_r3_mvit_trace $R/gpurun_out/prof_r3_trace
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r3_mvit_trace -o tr --output-format csv -- python $R/bench.py --arch mvit --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-side > $R/gpurun_out/r3_prof_mvit_trace.log 2>&1
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r3_trace -o tr --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-side > $R/gpurun_out/r3_prof_trace.log 2>&1
cd $R
python tools/timeline.py $(find gpurun_out/prof_r3_mvit_trace -name "*kernel_trace.csv" | head -1) > gpurun_out/r3_timeline_mvit.txt 2>&1
python tools/timeline.py $(find gpurun_out/prof_r3_trace -name "*kernel_trace.csv" | head -1) > gpurun_out/r3_timeline.txt 2>&1
find gpurun_out/prof_r3_mvit_trace gpurun_out/prof_r3_trace -name "*.csv" -size +20M -delete
du -sh gpurun_out/prof_r3*; head -12 gpurun_out/r3_timeline.txt
