#!/bin/bash
# round 4: what the dQ wave of the fused attention backward spends its block on (trace builds with parts removed)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4_attn; mkdir -p $O
for v in "" 1 2 4 8 12; do
  echo "--- ablate '${v:-0}'"
  PVRL_LIB_PATH=$PWD/procedurevrl_amd/csrc/variants/libpvrl_hip_fbt$v.so timeout 120 python tools/probe/attn_bwd_ab.py trace 2>&1 | grep -A5 "^wave 7" | grep -E "jb [234]"
  PVRL_LIB_PATH=$PWD/procedurevrl_amd/csrc/variants/libpvrl_hip_fbt$v.so timeout 120 python tools/probe/attn_bwd_ab.py trace 2>&1 | grep -A5 "^wave 0" | grep -E "jb [3]"
done | tee $O/dq_ablate.log
