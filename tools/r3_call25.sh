#!/bin/bash
# attn_win16: item pitch + 64 B (LDS bank conflicts): parity, then the segmenter bench (per-kernel times via rocprofv3 --stats)
TAG=${1:-r3w16}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_sam2_gpu.py -x -q -m gpu -k "attn or attention or hiera or Hiera or window" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o s -- python3 $GRAFT_REPO_ROOT/bench.py --workload sam2l --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; tail -1 $O/prof.log | cut -c1-200
[ -s $O/prof/s_kernel_stats.csv ] && grep -i "win16\|res64" $O/prof/s_kernel_stats.csv | cut -c1-160
find $O -name "*.db" -delete
