#!/bin/bash
# tools/gpurun_retry.sh TIMEOUT CMD -- gpurun with retries while the pod's GPU slots are busy (exit code 3: nothing charged)
T=$1; shift
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
