#!/bin/bash
# round 4, call b: per-phase s_memtime trace of the 8-wave NT kernel + PMC counters old vs new on one full-round shape
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r4_b; mkdir -p $O
timeout 600 python tools/probe/nt8_ab.py trace > $O/trace.txt 2>&1; echo "rc=$?" >> $O/trace.txt
cat $O/trace.txt | head -80
