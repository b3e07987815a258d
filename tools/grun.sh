#!/bin/bash
# retry wrapper around gpurun: exit code 3 = no slot free (nothing charged) -> wait and retry
# usage: tools/grun.sh <timeout_s> '<command>'
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
