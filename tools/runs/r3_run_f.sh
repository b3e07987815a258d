#!/bin/bash
python -m pytest tests -m gpu -q --durations=5 > gpurun_out/r3_pytest_f.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_f.log
BENCH_ARGS="--arch mvit" tools/profile_round.sh r3_mvit > gpurun_out/r3_prof_mvit.log 2>&1
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r3_mvit_trace -o tr --output-format csv -- python $R/bench.py --arch mvit --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-side > $R/gpurun_out/r3_prof_mvit_trace.log 2>&1
cd $R
python tools/timeline.py $(find gpurun_out/prof_r3_mvit_trace -name "*kernel_trace.csv" | head -1) > gpurun_out/r3_timeline_mvit.txt 2>&1
find gpurun_out/prof_r3_mvit_trace -name "*.csv" -size +20M -delete
tools/profile_round.sh r3 > gpurun_out/r3_prof_main.log 2>&1
tail -4 gpurun_out/r3_pytest_f.log; head -40 gpurun_out/r3_timeline_mvit.txt
