#!/bin/bash
# round 4: the fused spatial-attention backward with parts removed (variant builds -DPVRL_FB_ABLATE=<bits>; outputs garbage, time only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4_attn; mkdir -p $O
for v in "" fb1 fb2 fb3; do
  if [ -z "$v" ]; then unset PVRL_LIB_PATH; else export PVRL_LIB_PATH=$PWD/procedurevrl_amd/csrc/variants/libpvrl_hip_$v.so; fi
  echo "--- variant '${v:-full}'"
  PVRL_ATTN_BWD_FUSED=1 timeout 120 python tools/probe/attn_bwd_ab.py arm 2>&1 | grep "B="
done | tee $O/ablate.log
