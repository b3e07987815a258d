#!/bin/bash
# round 5: 8-rank launch test on one GPU (gloo), then the price of reserving CUs for RCCL (tools/probe/comm_cus_ab.py)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_comm_cus.txt; : > $O
timeout 1500 python -m pytest tests/test_launcher_gpu.py -m gpu -q -k "eight_ranks or spawns" 2>&1 | grep -v Warning | tail -8 >> $O
timeout 2400 python tools/probe/comm_cus_ab.py 2>&1 | tail -24 >> $O
cat $O
