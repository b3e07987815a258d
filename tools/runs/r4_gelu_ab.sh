#!/bin/bash
# GELU epilogue forms (csrc/common.h): VALU issue costs, the GELU / dGELU GEMMs with the product build next to the -DPVRL_GELU_FORM=0
# variant build (python tools/build_variant.py gelu0 --only gemm_nt.hip -DPVRL_GELU_FORM=0), and three interleaved pairs of the training
# step with either library.  Output: gpurun_out/r4_gelu_ab.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out; O=gpurun_out/r4_gelu_ab.txt
{
  hipcc --offload-arch=gfx950 -O3 -w -o /tmp/valu_rate tools/probe/valu_rate.hip && /tmp/valu_rate
  python tools/probe/nt8_ab.py gelu
  for i in 1 2 3; do
    for v in gelu0 product; do
      if [ $v = gelu0 ]; then export PVRL_LIB_PATH=$PWD/procedurevrl_amd/csrc/variants/libpvrl_hip_gelu0.so; else unset PVRL_LIB_PATH; fi
      python bench.py --steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v run $i:', d['value'], d['ms_per_step'])"
    done
  done
  unset PVRL_LIB_PATH
} > $O 2>&1
tail -12 $O
