#!/bin/bash
# round 4, call d: ablation of the 8-wave NT kernel's main loop + clock calibration of the traced workgroup
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r4_d; mkdir -p $O
timeout 900 python tools/probe/nt8_ab.py ablate > $O/ablate.txt 2>&1; echo "rc=$?" >> $O/ablate.txt
cat $O/ablate.txt


