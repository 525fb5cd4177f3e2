#!/bin/bash
TAG=${1:-r3t}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 300 python tools/probe/gemm_yardstick.py 2>&1 | tee $O/yardstick.txt
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/ys -o ys -- python3 $GRAFT_REPO_ROOT/tools/probe/gemm_yardstick.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; f=$(ls $O/ys/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cut -c1-200 $f | head -12
