#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
PVRL_RESID16=1 bash tools/profile_round.sh r6a > /dev/null 2>&1
PVRL_RESID16=0 bash tools/profile_round.sh r6a_off > /dev/null 2>&1
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for t in r6a r6a_off; do
  f=$(find gpurun_out/prof_$t/stats -name "*kernel_stats.csv" | head -1)
  python tools/summarize_prof.py $f gpurun_out/${t}_kernel_stats.csv "$t"
  head -30 gpurun_out/${t}_kernel_stats.csv
done
