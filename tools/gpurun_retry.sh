#!/bin/bash
# gpurun wrapper for a busy pool: retries ONLY when nothing ran (exit 3 = no box / slot free, nothing charged), at most 12 times.
# usage: tools/gpurun_retry.sh <timeout> '<command>'
T=$1; shift
for i in $(seq 1 12); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 100
done
exit 3
