#!/bin/bash
# same-box A/B of the training step for a variant build of the library: three interleaved pairs of
#   PVRL_LIB_PATH=procedurevrl_amd/csrc/variants/libpvrl_hip_<tag>.so | (product)  python bench.py --steps 20 --warmup 5 --no-side ...
# usage: bash tools/runs/r4_ab_lib.sh <tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:?tag}; O=gpurun_out/r4_ab_lib_$T.txt; : > $O
for i in 1 2 3; do
  for v in $T product; do
    if [ $v = product ]; then unset PVRL_LIB_PATH; else export PVRL_LIB_PATH=$PWD/procedurevrl_amd/csrc/variants/libpvrl_hip_$T.so; fi
    python bench.py --steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v run $i:', d['value'], d['ms_per_step'])" | tee -a $O
  done
done
unset PVRL_LIB_PATH
