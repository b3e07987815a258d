#!/bin/bash
# round 5: whole `-m gpu` suite, smoke, default bench (the driver's sequence)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_full.txt; : > $O
timeout 3300 python -X faulthandler -m pytest tests -m gpu -v 2>&1 | grep -v Warning > gpurun_out/r5_suite_v.txt
grep -n "FAILED\|ERROR" gpurun_out/r5_suite_v.txt | head -20 >> $O
grep -n -B2 -A12 "Fatal Python\|Segmentation" gpurun_out/r5_suite_v.txt | head -40 >> $O
tail -3 gpurun_out/r5_suite_v.txt >> $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 >> $O
( time python bench.py ) > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err; echo "bench rc=$?" >> $O
tail -4 gpurun_out/r5_bench_default.err >> $O
python - >> $O <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5_bench_default.json') if l.startswith('{')][-1])
for k in ("value","ms_per_step","dtype","parity","sustained","value_note","failed"):
    print(k, d.get(k))
print("cpu", d["cpu_baseline"].get("value"), d["cpu_baseline"].get("eval_leg"))
print("roofline", {k:v for k,v in d["roofline"].items() if k!="runner_up"})
print("runner_up", d["roofline"].get("runner_up"))
for s in d.get("side",[]): print("side", s.get("config","")[:60], s.get("value"), s.get("dtype"), s.get("parity"), s.get("error"))
PY
cat $O
# the two side configurations the default run leaves to --all-sides (configs[3], configs[4]) and the N > 1 machinery on one GPU
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --all-sides 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r5_bench_allsides.json
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-side --world1-rccl 2>/dev/null | grep "^{" | tail -1 > gpurun_out/r5_bench_world1.json
python - <<'PY' | tee -a gpurun_out/r5_suite3.txt
import json
d=json.loads(open('gpurun_out/r5_bench_allsides.json').read())
print("allsides headline", d["value"], d["ms_per_step"])
for s in d.get("side",[]): print("side", s.get("config","")[:60], s.get("value"), s.get("ms_per_step"), s.get("error"))
w=json.loads(open('gpurun_out/r5_bench_world1.json').read())
print("world1-rccl", w["value"], w["ms_per_step"])
PY
