#!/bin/bash
# usage: tools/gpurun_retry.sh <logfile> <timeout_s> '<command>'   - retries while gpurun reports "no box or slot free" (exit 3)
log=$1; to=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$to" -- "$@" > "$log" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "rc=$rc" >> "$log"; exit $rc; fi
  sleep 45
done
echo "rc=3 (gave up)" >> "$log"
