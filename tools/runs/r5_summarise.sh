#!/bin/bash
# HERE (no GPU): condense the raw rocprofv3 output that `r5_prof.sh` merged back under gpurun_out/ into the tracked profiles/r5_* files.
cd /root/repo
for t in "" _t32 _mvit; do
  d=gpurun_out/prof_r5$t
  [ -d $d ] || continue
  python tools/summarize_prof.py $(find $d/stats -name "*kernel_stats.csv" | head -1) profiles/r5${t}_kernel_stats.csv \
    "rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-side ${t:+($t)} ; round-5 final code, fp16 operands (default library), one stream"
  ( cd tools && python summarize_pmc.py ../$d/FETCH_SIZE/pmc_counter_collection.csv ../$d/WRITE_SIZE/pmc_counter_collection.csv \
      ../$d/SQ_VALU_MFMA_BUSY_CYCLES/pmc_counter_collection.csv ../profiles/r5${t}_pmc_hbm_mfma.csv ../profiles/r5${t}_traffic.json )
done
cp gpurun_out/r5_timeline.txt profiles/r5_timeline.txt
cp gpurun_out/r5_timeline_full.txt profiles/r5_timeline_full.txt
head -12 profiles/r5_pmc_hbm_mfma.csv
