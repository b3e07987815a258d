#!/bin/bash
for t in 22 42 26; do
PVRL_NT_TILE=$t python tools/probe/mvit_gemm_times.py > gpurun_out/r3_w_shapes_t$t.txt 2>&1
done
python tools/probe/mvit_gemm_times.py > gpurun_out/r3_w_shapes_def.txt 2>&1
tail -n 1 gpurun_out/r3_w_shapes_*.txt
