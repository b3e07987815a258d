#!/bin/bash
# round 4, call a: first run of the persistent 8-wave NT kernel -- bit-compare vs the 16-wave kernel, race screen, timing; kernel suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4_a; mkdir -p $O
timeout 1200 python tools/probe/nt8_ab.py check race 10 time > $O/nt8_ab.txt 2>&1; echo "rc=$?" >> $O/nt8_ab.txt
tail -40 $O/nt8_ab.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q > $O/kernels.txt 2>&1; echo "rc=$?" >> $O/kernels.txt
tail -5 $O/kernels.txt
