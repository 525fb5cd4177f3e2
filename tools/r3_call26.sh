#!/bin/bash
# refinement head with packed-f32 FMAs: parity (golden fixtures + SAM tests), per-kernel time
TAG=${1:-r3rf}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_sam2_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "refine or upsample or golden or wrapper or boundary or hiera_l" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o s -- python3 $GRAFT_REPO_ROOT/bench.py --workload sam2l --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
[ -s $O/prof/s_kernel_stats.csv ] && grep -i "upsample_refine\|bilinear\|hyper_mask" $O/prof/s_kernel_stats.csv | cut -c1-200
find $O -name "*.db" -delete
echo done
