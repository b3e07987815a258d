#!/bin/bash
# end-of-round profiles: rocprofv3 stats + the three separate PMC passes (tools/profile_round.sh) for configs[1], configs[3] (T = 32) and
# configs[4] (MViTv2-S), the kernel-trace timeline of one step of configs[1] and of the full pre-training step; summaries written
# on the box into gpurun_out/r6*_{kernel_stats,pmc_hbm_mfma}.csv / r6*_traffic.json / r6_timeline*.txt (copied to profiles/ afterwards).
# usage (on the GPU box, via gpurun): bash tools/runs/r6_prof.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
rm -rf gpurun_out/prof_r6*
tools/profile_round.sh r6 > gpurun_out/r6_prof_main.log 2>&1
cd $R; BENCH_ARGS="--arch mvit" tools/profile_round.sh r6_mvit > gpurun_out/r6_prof_mvit.log 2>&1
cd $R; BENCH_ARGS="--frames 32 --batch 8" tools/profile_round.sh r6_t32 > gpurun_out/r6_prof_t32.log 2>&1
cd $R
for t in "" _mvit _t32; do
  d=gpurun_out/prof_r6$t
  python tools/summarize_prof.py $(find $d/stats -name "*kernel_stats.csv" | head -1) gpurun_out/r6${t}_kernel_stats.csv \
    "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-side ${t:+($t)} ; round-6 final code, fp16 operands (default library), split residual stream, one stream"
  ( cd tools && python summarize_pmc.py ../$d/FETCH_SIZE/pmc_counter_collection.csv ../$d/WRITE_SIZE/pmc_counter_collection.csv \
      ../$d/SQ_VALU_MFMA_BUSY_CYCLES/pmc_counter_collection.csv ../gpurun_out/r6${t}_pmc_hbm_mfma.csv ../gpurun_out/r6${t}_traffic.json )
done
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r6_trace -o tr --output-format csv -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-side --no-parity-probe > $R/gpurun_out/r6_prof_trace.log 2>&1
rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r6_trace_full -o tr --output-format csv -- python $R/tools/bench_full_step.py --steps 4 --warmup 6 > $R/gpurun_out/r6_prof_trace_full.log 2>&1
cd $R
python tools/timeline.py $(find gpurun_out/prof_r6_trace -name "*kernel_trace.csv" | head -1) > gpurun_out/r6_timeline.txt 2>&1
python tools/timeline.py $(find gpurun_out/prof_r6_trace_full -name "*kernel_trace.csv" | head -1) > gpurun_out/r6_timeline_full.txt 2>&1
rm -rf gpurun_out/prof_r6*
head -14 gpurun_out/r6_pmc_hbm_mfma.csv; head -12 gpurun_out/r6_timeline.txt; head -8 gpurun_out/r6_timeline_full.txt; head -12 gpurun_out/r6_mvit_pmc_hbm_mfma.csv | cut -c1-150
