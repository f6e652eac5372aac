#!/bin/bash
# gpurun with retries while no slot / box is free (exit code 3): tools/gpr.sh <timeout s> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 45
done
exit 3
