#!/bin/bash
# reproduce the intermittent abort of tests/test_train_loop_gpu.py::test_train_checkpoint_resume: N isolated runs, stderr kept
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_flake.txt; : > $O
N=${1:-12}
for i in $(seq 1 $N); do
  timeout 300 python -X faulthandler -m pytest tests/test_train_loop_gpu.py -m gpu -q -x > /tmp/flake_$i.log 2>&1
  rc=$?
  echo "run $i rc=$rc $(grep -E 'passed|failed' /tmp/flake_$i.log | tail -1)" >> $O
  if [ $rc -ne 0 ]; then echo "---- log of run $i" >> $O; head -c 6000 /tmp/flake_$i.log >> $O; fi
done
cat $O | cut -c1-300 | head -80
