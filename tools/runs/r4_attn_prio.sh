#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4_attn; mkdir -p $O
for v in fbtp00 fbtp0 fbtp; do
  echo "--- $v"
  PVRL_LIB_PATH=$PWD/procedurevrl_amd/csrc/variants/libpvrl_hip_$v.so timeout 120 python tools/probe/attn_bwd_ab.py trace 2>&1 > $O/trace_$v.log
  grep -A5 "^wave 7" $O/trace_$v.log | grep -E "wave|jb [34]"; grep -A5 "^wave 0" $O/trace_$v.log | grep -E "jb [3]";  grep -A5 "^wave 4" $O/trace_$v.log | grep -E "jb [3]"
  PVRL_LIB_PATH=$PWD/procedurevrl_amd/csrc/variants/libpvrl_hip_$v.so timeout 120 python tools/probe/attn_bwd_ab.py arm 2>&1 | grep "B="
done | tee $O/prio.log
