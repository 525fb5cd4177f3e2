#!/bin/bash
# staged O stores in the head_dim-72 attention kernels: parity, then A/B on the segmenter bench
TAG=${1:-r3s}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_ops_gpu.py tests/test_sam2_gpu.py -x -q -m gpu -k "attn or attention or hiera or Hiera or window or global" > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for os in 1 0 1 0; do
  CVMI_ATTN_OSTAGE=$os timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_os$os.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/sam_os$os.json").read().strip().splitlines()[-1])
print("OSTAGE=$os:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:34], t["us_per_launch"]) for t in d["top_launches"] if "attn" in t["kernel"]])
PY
done
