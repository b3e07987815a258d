#!/bin/bash
# round 4: the full `-m gpu` suite + smoke + the default bench (what the driver runs at round end), outputs under gpurun_out/r4_full_<tag>
cd "${GRAFT_REPO_ROOT:-/root/repo}"
T=${1:-x}; O=gpurun_out/r4_full_$T; mkdir -p $O
S=$(date +%s)
python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$? ($(( $(date +%s) - S )) s)" >> $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log
S=$(date +%s)
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$? ($(( $(date +%s) - S )) s)" >> $O/bench.err
tail -n 14 $O/pytest.log; tail -n 3 $O/smoke.log; tail -n 2 $O/bench.err; cut -c1-2500 $O/bench.json
