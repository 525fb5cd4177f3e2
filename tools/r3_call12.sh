#!/bin/bash
TAG=${1:-r3m}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for ns in 1 2 3; do
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr$ns -o t -- python3 bench.py --streams $ns --no-cpu-baseline --no-profile-pass --steps 4 --warmup 2 > $O/tr$ns.log 2>&1
  echo "streams=$ns: $(tail -1 $O/tr$ns.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  python3 tools/overlap_stats.py $(find $O/tr$ns -name "*kernel_trace.csv" | head -1) 0.45
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +20M -delete
