#!/bin/bash
# round 4, call c: the (row half, k half) schedule of the 8-wave NT kernel: bit-compare, race screen, timing, per-phase trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r4_c; mkdir -p $O
timeout 1200 python tools/probe/nt8_ab.py check race 10 time > $O/nt8_ab.txt 2>&1; echo "rc=$?" >> $O/nt8_ab.txt
grep -v "^checked\|^race" $O/nt8_ab.txt | tail -20
timeout 600 python tools/probe/nt8_ab.py trace > $O/trace.txt 2>&1; echo "rc=$?" >> $O/trace.txt
head -24 $O/trace.txt
