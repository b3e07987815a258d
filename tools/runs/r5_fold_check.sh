#!/bin/bash
# round 5: 1/S and the non-finite check folded into the gradient-writing kernels -- GPU suite in the fp16 flavour, then clips/s of both flavours
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_fold_check.txt; : > $O
PVRL_OPERAND=f16 timeout 2400 python -m pytest tests -m gpu -x -q --deselect tests/test_f16_flavour_gpu.py 2>&1 | tail -15 >> $O
for f in f16 bf16; do
  echo "== $f" >> $O
  PVRL_OPERAND=$f timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing --parity-probe 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('parity'))" >> $O
done
cat $O
