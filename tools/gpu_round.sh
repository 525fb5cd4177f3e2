#!/bin/bash
# One gpurun call: box description, the GPU test suite, the default bench line.  Output under gpurun_out/<tag>/.
set -o pipefail
TAG=${1:-r02a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
{ nproc; python -c "import os;print('affinity',len(os.sched_getaffinity(0)))"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; grep -m1 "model name" /proc/cpuinfo; free -g | head -2; } > $O/box.txt 2>&1
timeout -k 10 1000 python -m pytest tests -m gpu -q -rA --durations=15 > $O/pytest.log 2>&1; rc=$?
tail -5 $O/pytest.log
echo "pytest rc=$rc"
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py > $O/bench_circuit.json 2> $O/bench_circuit.err; rc=$?
echo "bench rc=$rc"; tail -c 1500 $O/bench_circuit.json
exit $rc
