#!/bin/bash
# One gpurun call: box description, the GPU test suite, the default bench line, optional extra commands ($2..).
# Output under gpurun_out/<tag>/.
set -o pipefail
TAG=${1:-r02a}; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
{ nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep -m1 "model name" /proc/cpuinfo; } > $O/box.txt 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -q -rA --durations=15 > $O/pytest.log 2>&1; rc=$?
grep -E "^FAILED|^ERROR| passed| failed" $O/pytest.log | tail -15
echo "pytest rc=$rc"
if [ "$SKIP_BENCH" != "1" ]; then
  timeout -k 10 500 python bench.py > $O/bench_circuit.json 2> $O/bench_circuit.err; brc=$?
  echo "bench rc=$brc"; tail -c 600 $O/bench_circuit.json
fi
i=0
for cmd in "$@"; do
  i=$((i+1)); echo "== extra $i: $cmd"
  timeout -k 10 400 bash -c "$cmd" > $O/extra$i.out 2> $O/extra$i.err; echo "rc=$?"; tail -c 400 $O/extra$i.out
done
exit $rc
