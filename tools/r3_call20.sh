#!/bin/bash
# 16 x 16 windows: resident-K/V kernel (attn_res256) vs the streaming kernel (attn_dma72, 64-key tiles double-buffered)
TAG=${1:-r3w}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for rs in 2 0 2 0; do
  CVMI_ATTN_RES256=$rs timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_rs$rs.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/sam_rs$rs.json").read().strip().splitlines()[-1])
print("RES256=$rs:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:34], t["launches_per_pass"], t["us_per_launch"], t["bound"], t["frac"]) for t in d["top_launches"] if "attn" in t["kernel"]])
PY
done
