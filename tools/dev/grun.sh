#!/bin/bash
# usage: tools/dev/grun.sh <timeout-seconds> '<command>'   -- gpurun with retries while no GPU slot is free (exit code 3)
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
