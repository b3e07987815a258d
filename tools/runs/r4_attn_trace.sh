#!/bin/bash
# round 4: fused attention backward: parity check, cycle-stamp trace (variant build -DPVRL_FB_TRACE=1), A/B timing
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4_attn; mkdir -p $O
T=${1:-x}
timeout 300 python tools/probe/attn_bwd_ab.py check 2>&1 | grep -E "BAD|CHECK|Error|error" | head
PVRL_LIB_PATH=$PWD/procedurevrl_amd/csrc/variants/libpvrl_hip_fbt.so timeout 120 python tools/probe/attn_bwd_ab.py trace > $O/trace_$T.log 2>&1
grep -E "^wave|jb 3|jb 0" $O/trace_$T.log
timeout 200 python tools/probe/attn_bwd_ab.py time 2>&1 | grep -E "^---|B=" | tee $O/time_$T.log
