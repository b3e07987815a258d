#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_lnbench.txt; : > $O
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "split or layernorm" 2>&1 | tail -5 >> $O
python tools/probe/resid16_bench.py >> $O 2>&1
for n in 384 768 1024; do PVRL_LN_BWD_BLOCKS=$n python tools/probe/resid16_bench.py >> $O 2>&1; done
cat $O
