#!/bin/bash
# hypothesis: the intermittent abort / hang of the train-loop tests depends on WHICH pool streams (-> hardware queues) the loop's side
# streams get, i.e. on how many streams earlier tests created.  k dummy streams first, then the test, each k in a fresh process.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6_flake3.txt; : > $O
for k in $(seq 0 3 45); do
  timeout 120 python -X faulthandler -c "
import sys, torch, pytest
keep = [torch.cuda.Stream() for _ in range($k)]
sys.exit(pytest.main(['tests/test_train_loop_gpu.py', '-m', 'gpu', '-q', '-x', '-k', 'checkpoint_resume or loss_decreases or eval_epoch']))" > /tmp/k_$k.log 2>&1
  echo "k=$k rc=$? $(grep -E ' passed| failed|Fatal' /tmp/k_$k.log | tail -1)" >> $O
done
cat $O
