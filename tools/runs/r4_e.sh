#!/bin/bash
# round 4, call e: step-level A/B of the 8-wave NT kernel (PVRL_NT8=1) vs the 16-wave one (PVRL_NT8=0), interleaved on one box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=$PWD/gpurun_out/r4_e; mkdir -p $O
for i in 1 2 3; do
  for v in 0 1; do
    PVRL_NT8=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing > $O/b_${v}_$i.json 2> $O/b_${v}_$i.err
    python - <<PY
import json
try:
    d=json.loads(open("$O/b_${v}_$i.json").read().strip().splitlines()[-1]); print("NT8=$v run $i:", d["value"], d["ms_per_step"])
except Exception as e: print("NT8=$v run $i failed", e)
PY
  done
done
