#!/bin/bash
# row-block-persistent fc2 (gemm256x192r_kernel): parity, then A/B on the segmenter bench
TAG=${1:-r3u}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_sam2_gpu.py -x -q -m gpu -k "row_statistics or hiera or Hiera or gemm or conv" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for ps in 1 0 1 0; do
  CVMI_G192_PERSIST=$ps timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_ps$ps.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/sam_ps$ps.json").read().strip().splitlines()[-1])
print("PERSIST=$ps:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:34], t["us_per_launch"], t["frac"]) for t in d["top_launches"] if "gemm256x192" in t["kernel"]])
PY
done
