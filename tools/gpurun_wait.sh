#!/bin/bash
# gpurun with patience: exit code 3 = no box / slot free, nothing ran and nothing was charged -> wait and ask again (never re-runs a command
# that did run: any other exit code is returned as it is).  usage: tools/gpurun_wait.sh <timeout-seconds> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
exit 3
