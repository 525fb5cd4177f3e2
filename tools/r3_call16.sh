#!/bin/bash
TAG=${1:-r3r}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for dg in 0 1 2 3; do
  CVMI_ATTN_DIAG=$dg timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 3 > $O/sam_diag$dg.json 2>/dev/null
  python3 - <<PY
import json
d=json.loads(open("$O/sam_diag$dg.json").read().strip().splitlines()[-1])
print("DIAG=$dg:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:30], t["us_per_launch"]) for t in d["top_launches"] if "attn_res256" in t["kernel"]])
PY
done
