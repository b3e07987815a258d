#!/bin/bash
# usage: r6_prof.sh <tag>; rocprofv3 stats + PMC passes over the bench command, summaries into gpurun_out/
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=$1
bash tools/profile_round.sh $TAG > /dev/null 2>&1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
f=$(find gpurun_out/prof_$TAG/stats -name "*kernel_stats.csv" | head -1)
python tools/summarize_prof.py $f gpurun_out/${TAG}_kernel_stats.csv "$TAG"
P=gpurun_out/prof_$TAG
python tools/summarize_pmc.py $(find $P/FETCH_SIZE -name "*counter_collection.csv") $(find $P/WRITE_SIZE -name "*counter_collection.csv") \
   $(find $P/SQ_VALU_MFMA_BUSY_CYCLES -name "*counter_collection.csv") gpurun_out/${TAG}_pmc_hbm_mfma.csv gpurun_out/${TAG}_traffic.json 2>&1 | tail -3
head -26 gpurun_out/${TAG}_pmc_hbm_mfma.csv
rm -rf $P
