#!/bin/bash
S=$(date +%s)
python -m pytest tests -m gpu -q --durations=8 > gpurun_out/r3_pytest_full.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> gpurun_out/r3_pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r3_smoke_full.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r3_smoke_full.log
S=$(date +%s)
python bench.py > gpurun_out/r3_bench_final.json 2> gpurun_out/r3_bench_final.err; echo "bench rc=$? ($(( $(date +%s) - S )) s)" >> gpurun_out/r3_bench_final.err
tail -n 14 gpurun_out/r3_pytest_full.log; tail -n 3 gpurun_out/r3_smoke_full.log; tail -n 2 gpurun_out/r3_bench_final.err; cat gpurun_out/r3_bench_final.json | cut -c1-3000
