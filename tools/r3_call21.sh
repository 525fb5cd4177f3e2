#!/bin/bash
# global attention: 4-wave workgroups (4 per CU) vs 8-wave (1 per CU)
TAG=${1:-r3x}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_sam2_gpu.py -x -q -m gpu -k "attn or attention or hiera or Hiera or window or global" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for nw in 8; do
  CVMI_ATTN_DMA72_NW=$nw timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_nw$nw.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/sam_nw$nw.json").read().strip().splitlines()[-1])
print("NW=$nw:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:34], t["launches_per_pass"], t["us_per_launch"], t["bound"], t["frac"]) for t in d["top_launches"] if "attn" in t["kernel"]])
PY
done
