#!/bin/bash
python -m pytest tests -m gpu -q --durations=6 > gpurun_out/r3_pytest_d.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_d.log
for i in 1 2; do
  PVRL_HEAD_ENGINE=0 python tools/bench_full_step.py --steps 10 > gpurun_out/r3_full_autograd_$i.json 2>/dev/null
  python tools/bench_full_step.py --steps 10 > gpurun_out/r3_full_engine_$i.json 2>/dev/null
done
python tools/bench_full_step.py --arch mvit --steps 10 > gpurun_out/r3_full_engine_mvit.json 2>/dev/null
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r3_full -o full --output-format csv -- python $R/tools/bench_full_step.py --steps 4 > $R/gpurun_out/r3_prof_full.log 2>&1
cd $R
python tools/timeline.py $(find gpurun_out/prof_r3_full -name "*kernel_trace.csv" | head -1) > gpurun_out/r3_timeline_full.txt 2>&1
find gpurun_out/prof_r3_full -name "*.csv" -size +20M -delete
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
python bench.py $B > gpurun_out/r3_d_bench1.json 2>/dev/null
python bench.py $B > gpurun_out/r3_d_bench2.json 2>/dev/null
tail -5 gpurun_out/r3_pytest_d.log; grep -h -o '"value": [0-9.]*' gpurun_out/r3_full_*.json gpurun_out/r3_d_bench*.json; head -12 gpurun_out/r3_timeline_full.txt
