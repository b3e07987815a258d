#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace/stats + three separate PMC passes over the bench command,
# raw output under gpurun_out/prof_<tag>/; summarise afterwards with tools/summarize_prof.py / summarize_pmc.py.
# usage: [BENCH_ARGS="--frames 32 --batch 8"] tools/profile_round.sh <tag>
TAG=${1:-r1}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $PWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-side $BENCH_ARGS"
cd /tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- $CMD > "$OUT/stats.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_VALU_MFMA_BUSY_CYCLES; do
  rocprofv3 --kernel-trace --pmc $c -d "$OUT/$c" -o pmc --output-format csv -- $CMD > "$OUT/$c.log" 2>&1
done
cd "$OUT"
find . -name "*.csv" | head -20
# keep only what the summarisers need (the merge-back limit is 64 MiB)
find . -name "*kernel_trace.csv" -path "./stats/*" -delete
du -sh .
