#!/bin/bash
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k gemm > gpurun_out/r3_pytest_x.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r3_pytest_x.log
python tools/probe/mvit_pool_times.py > gpurun_out/r3_x_pool.txt 2>&1
B="--steps 20 --warmup 5 --no-side --no-cpu-baseline --no-kernel-timing"
python bench.py $B --arch mvit > gpurun_out/r3_x_mvit_1.json 2>/dev/null
tail -n 3 gpurun_out/r3_pytest_x.log; grep -H -o '"value": [0-9.]*' gpurun_out/r3_x_*.json; cat gpurun_out/r3_x_pool.txt
