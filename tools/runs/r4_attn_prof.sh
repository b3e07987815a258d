#!/bin/bash
# round 4: kernel trace of the spatial-attention A/B arm (fused backward on)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r4_attn; mkdir -p $O
export TMPDIR=/tmp
PVRL_ATTN_BWD_FUSED=${1:-1} timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$1 -o t -- python tools/probe/attn_bwd_ab.py arm > $O/prof_$1.log 2>&1
f=$(find $O/prof_$1 -name '*kernel_stats.csv' | head -1)
head -12 "$f" | cut -c1-200
