#!/bin/bash
# round 5: the driver's sequence on one box -- `pytest -m gpu`, smoke(), the default `python bench.py`
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5_suite.txt; : > $O
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | grep -v Warning | tail -25 >> $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -12 >> $O
( time python bench.py ) > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err; echo "bench rc=$?" >> $O
tail -5 gpurun_out/r5_bench_default.err >> $O
python - >> $O <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5_bench_default.json') if l.startswith('{')][-1])
for k in ("value","ms_per_step","dtype","parity","sustained","value_note","cpu_baseline","failed"):
    print(k, d.get(k))
print("roofline", {k:v for k,v in d["roofline"].items() if k!="runner_up"})
for s in d.get("side",[]): print("side", s.get("config","")[:60], s.get("value"), s.get("dtype"), s.get("parity"), s.get("error"))
PY
cat $O
