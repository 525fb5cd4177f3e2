#!/bin/bash
# K = 576 token-stationary linears: 16x16x32 (tok_linear16.hip) vs 32x32x16 (tok_linear.hip) MFMA shape, end of round 3
TAG=${1:-r3m16}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for m in 1 0 1 0; do
  CVMI_TOKLIN_M16=$m timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_m$m.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/sam_m$m.json").read().strip().splitlines()[-1])
print("M16=$m:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:48], t["us_per_launch"]) for t in d["top_launches"] if "tok_linear" in t["kernel"]])
PY
done
