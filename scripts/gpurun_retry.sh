#!/bin/bash
# usage: scripts/gpurun_retry.sh <timeout-seconds> '<command>'   -- retries while the pod has no free GPU slot
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@" > /tmp/gpurun_last.log 2>&1
  rc=$?
  if grep -q "status=transient" /tmp/gpurun_last.log; then sleep 90; continue; fi
  break
done
cat /tmp/gpurun_last.log | tail -80
exit $rc
