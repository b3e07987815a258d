#!/bin/bash
# round 6: split residual stream (PVRL_RESID16): kernel + e2e parity, then the step with it on / off (three interleaved pairs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_resid16.txt; : > $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "split or layernorm or gemm_nt" 2>&1 | tail -5 >> $O
timeout 2400 python -m pytest tests/test_e2e_gpu.py -m gpu -q 2>&1 | tail -5 >> $O
for i in 1 2 3; do for m in 1 0; do
  echo -n "PVRL_RESID16=$m : " >> $O
  PVRL_RESID16=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing 2>/dev/null | grep "^{" | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('loss'), d.get('parity'))" >> $O
done; done
cat $O
