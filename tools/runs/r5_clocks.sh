#!/bin/bash
# round 5: shader clock / power while the training step runs, per operand flavour (is the fp16 flavour's 3 % a clock effect?)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_clocks.txt; : > $O
for f in f16 bf16 f16 bf16; do
  echo "== PVRL_OPERAND=$f" >> $O
  PVRL_OPERAND=$f python bench.py --steps 400 --warmup 5 --no-cpu-baseline --no-side --no-kernel-timing > /tmp/b_$f.json 2>/dev/null &
  BP=$!
  sleep 22
  for i in 1 2 3 4 5 6; do
    /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)\|Average Graphics\|Socket" | tr -s ' ' | cut -c1-110 >> $O
    sleep 1.5
  done
  wait $BP
  grep "^{" /tmp/b_$f.json | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('clips/s', d['value'], 'ms', d['ms_per_step'])" >> $O
done
cat $O
