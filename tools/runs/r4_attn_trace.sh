#!/bin/bash
# round 4: cycle-stamp trace of the fused attention backward (variant build -DPVRL_FB_TRACE=1) + check + A/B timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4_attn; mkdir -p $O
T=${1:-x}
timeout 300 python tools/probe/attn_bwd_ab.py check 2>&1 | grep -E "BAD|CHECK"
PVRL_LIB_PATH=$PWD/procedurevrl_amd/csrc/variants/libpvrl_hip_fbt.so timeout 120 python tools/probe/attn_bwd_ab.py trace > $O/trace_$T.log 2>&1
grep -E "^wave|jb 3" $O/trace_$T.log
for v in "" fb2; do
  if [ -z "$v" ]; then unset PVRL_LIB_PATH; else export PVRL_LIB_PATH=$PWD/procedurevrl_amd/csrc/variants/libpvrl_hip_$v.so; fi
  echo "--- variant '${v:-full}'"
  PVRL_ATTN_BWD_FUSED=1 timeout 120 python tools/probe/attn_bwd_ab.py arm 2>&1 | grep "B="
done | tee $O/ablate_$T.log
