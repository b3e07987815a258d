cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r4_ab_gm_side.txt; : > $O
for cfg in "--arch mvit" "--frames 32 --batch 8" ""; do
 for i in 1 2; do
  for v in gmold product; do
    if [ $v = product ]; then unset PVRL_LIB_PATH; else export PVRL_LIB_PATH=$PWD/procedurevrl_amd/csrc/variants/libpvrl_hip_gmold.so; fi
    python bench.py $cfg --steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$cfg] $v run $i:', d['value'], d['ms_per_step'])" | tee -a $O
  done
 done
done
unset PVRL_LIB_PATH
python tools/probe/run_check.py check_gemm_nt 2>&1 | grep -c "^ok"; python tools/probe/run_check.py check_gemm_nt 2>&1 | grep "^BAD"
