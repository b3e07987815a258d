#!/bin/bash
# round 5: whole `-m gpu` suite (no -x, verbose log kept), then smoke
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 3300 python -X faulthandler -m pytest tests -m gpu -v 2>&1 | grep -v Warning > gpurun_out/r5_suite_v.txt
grep -n "PASSED\|FAILED\|ERROR\|SKIPPED" gpurun_out/r5_suite_v.txt | grep -v PASSED | head -30
grep -c PASSED gpurun_out/r5_suite_v.txt
grep -n -B2 -A12 "Fatal Python\|Segmentation" gpurun_out/r5_suite_v.txt | head -60
tail -5 gpurun_out/r5_suite_v.txt
