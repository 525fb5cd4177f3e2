#!/bin/bash
TAG=${1:-r3q}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for r in 1 2; do for fm in 1 0; do
  CVMI_SAM_FUSED_MLP=$fm timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_fm${fm}_$r.json 2>/dev/null
  python3 - <<PY
import json
d=json.loads(open("$O/sam_fm${fm}_$r.json").read().strip().splitlines()[-1])
print("round $r FUSED_MLP=$fm:", d["ms_per_step"], "ms/step;", {k:v["ms"] for k,v in d["stages"]["sam2l"]["breakdown"].items() if k in ("gemm","mlp_fused","layernorm")})
PY
done; done
