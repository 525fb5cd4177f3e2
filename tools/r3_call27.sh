#!/bin/bash
# what bounds the persistent fc2 K-loop: timing-only builds of the same kernel (CVMI_G192_DIAG; results are wrong in every mode but 0)
TAG=${1:-r3fd}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for dg in 0 2 0; do
  CVMI_G192_DIAG=$dg timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 3 > $O/sam_dg$dg.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/sam_dg$dg.json").read().strip().splitlines()[-1])
print("DIAG=$dg:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:30], t["us_per_launch"]) for t in d["top_launches"] if "gemm256x192r" in t["kernel"]])
PY
done
