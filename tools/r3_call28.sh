#!/bin/bash
TAG=${1:-r3ff}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_sam2_gpu.py -x -q -m gpu -k "row_statistics or permutation or gemm or conv or hiera_l" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
