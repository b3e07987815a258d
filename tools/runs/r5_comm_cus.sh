#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_comm_cus.txt; : > $O
for rep in 1 2; do for q in 4 8; do
  echo "== GPU_MAX_HW_QUEUES=$q (pass $rep)" >> $O
  COMM_CUS_CASES="0:1,1:1,1:0,0:0" GPU_MAX_HW_QUEUES=$q timeout 1200 python tools/probe/comm_cus_ab.py 2>&1 | grep "^{" >> $O
done; done
cat $O
