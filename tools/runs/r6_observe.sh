#!/bin/bash
# every parity check with its observed error (for setting bounds), then the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 2400 python tools/gpu_check.py check_text_tower check_forecast check_mvit_timed check_mvit_s_full check_split check_block_golden > gpurun_out/r6_observe.txt 2>&1
tail -80 gpurun_out/r6_observe.txt | grep -v "^#" | cut -c1-200
cat gpurun_out/r3_mvit_timed_obs_f16.txt
timeout 1700 python bench.py > gpurun_out/r6_bench_default.json 2> gpurun_out/r6_bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/r6_bench_default.json") if l.startswith("{")][-1])
print(d["value"], d["ms_per_step"], d["dtype"], d.get("parity"), d.get("failed"))
print(json.dumps(d.get("roofline"))[:600])
print(json.dumps(d.get("cpu_baseline"))[:900])
print(d.get("sustained"))
for s in d.get("side", []): print(s.get("config","")[:60], s.get("value"), s.get("ms_per_step"), s.get("parity"), s.get("error"))
PY
