#!/bin/bash
# Round 6: with the MViT tests in front, the train-loop tests aborted / hung inside the HIP runtime in 3 of 4 runs -- dead models'
# captured graphs and pools torn down by the garbage collector during the next test's steps.  tests/conftest.py now releases every GPU
# test's objects at a quiet point (5 of 5 pass); PVRL_TEST_NO_GC_FIXTURE=1 switches that off (this script's first argument "off").
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_flake4.txt; : > $O
ulimit -c 0
[ "$1" = off ] && export PVRL_TEST_NO_GC_FIXTURE=1
for i in 1 2 3 4 5; do
  S=$(date +%s)
  timeout 200 python -X faulthandler -m pytest tests/test_mvit_gpu.py tests/test_train_loop_gpu.py -m gpu -q -x -p no:cacheprovider > /tmp/c_$i.log 2>&1
  echo "run $i rc=$? $(( $(date +%s) - S )) s $(grep -E ' passed| failed|Fatal' /tmp/c_$i.log | tail -1)" >> $O
done
cat $O
