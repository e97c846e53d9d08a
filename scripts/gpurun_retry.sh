#!/bin/bash
# build container: `gpurun` with retries while every GPU slot of the pod is busy (exit code 3: nothing charged).  Usage: scripts/gpurun_retry.sh <timeout s> <log> '<command>'
T=$1; LOG=$2; shift 2
for i in $(seq 1 30); do
  gpurun --timeout "$T" -- "$@" > "$LOG" 2>&1; rc=$?
  if ! grep -q "status=transient" "$LOG"; then exit $rc; fi
  sleep 90
done
exit 3
