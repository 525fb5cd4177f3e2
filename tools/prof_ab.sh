#!/bin/bash
# Per-kernel A/B of one environment variable under rocprofv3 --kernel-trace --stats (two separate runs in one gpurun call).
# usage: tools/prof_ab.sh <tag> "<bench args>" VAR=a VAR=b
TAG=$1; ARGS=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$TAG
for v in "$@"; do
  export $v
  d=gpurun_out/$TAG/$(echo $v | tr '=' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python bench.py $ARGS --steps 5 --warmup 2 --no-cpu-baseline > $d.json 2> $d.err || exit 1
  python tools/prof_summary.py $d $d.md "$v" && rm -rf $d
done
