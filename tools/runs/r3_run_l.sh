#!/bin/bash
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
for m in 0 0x7f 0x20 0x01 0x21 0x08 0x28 0x29 0x02 0x77 0 0x7f; do
  PVRL_NT_PERSIST=$m timeout 300 python bench.py $B > gpurun_out/r3_l_$m.json 2>/dev/null
  echo "mask $m: $(grep -o '"value": [0-9.]*' gpurun_out/r3_l_$m.json)"
done
