#!/bin/bash
# round 5: the cls rows' fp32 chain (csrc/cls_chain.hip) -- kernel check, parity of the 2-clip full-size step and clips/s with the chain
# on / off in both operand flavours.  Output: gpurun_out/r5_cls_ab.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_cls_ab.txt; : > $O
for f in f16 bf16; do
  echo "== kernel check $f" >> $O
  PVRL_OPERAND=$f timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "cls_linear or gemm_f32" 2>&1 | tail -5 >> $O
  for c in 1 0; do
    echo "== $f PVRL_CLS_FP32=$c" >> $O
    PVRL_OPERAND=$f PVRL_CLS_FP32=$c timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing --parity-probe 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d.get('parity'))" >> $O
  done
done
cat $O
