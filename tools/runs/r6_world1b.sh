#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_world1.txt; : > $O
timeout 900 python -m pytest tests/test_nccl_single_rank_gpu.py tests/test_two_rank_gloo_gpu.py -m gpu -q 2>&1 | tail -3 >> $O
for i in 1 2; do for m in 1 0; do
  echo "PVRL_HOOK_TAIL_SPLIT=$m --world1-rccl:" >> $O
  PVRL_HOOK_TAIL_SPLIT=$m timeout 600 python bench.py --world1-rccl --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing --no-parity-probe 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['comm']; print(' ', d['value'], d['ms_per_step'], 'exposed', c['exposed_ms_per_step'], 'per chunk', c['allreduce_ms_per_chunk'], c['chunk_mb'])" >> $O
done; done
echo "PVRL_GRAD_COMM=bf16 --world1-rccl:" >> $O
PVRL_GRAD_COMM=bf16 timeout 600 python bench.py --world1-rccl --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing --no-parity-probe 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['comm']; print(' ', d['value'], d['ms_per_step'], 'exposed', c['exposed_ms_per_step'], 'per chunk', c['allreduce_ms_per_chunk'], c['chunk_mb'])" >> $O
echo "no data-parallel machinery:" >> $O
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing --no-parity-probe 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(' ', d['value'], d['ms_per_step'])" >> $O
cat $O
