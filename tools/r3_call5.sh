#!/bin/bash
TAG=${1:-r3e}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for m in 1 0; do
  CVMI_G192_M16=$m timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "gemm_256_tile or row_statistics" > $O/pytest_m16_$m.log 2>&1; echo "pytest M16=$m rc=$?"; tail -2 $O/pytest_m16_$m.log
done
for r in 1 2 3; do for m in 1 0; do
  CVMI_G192_M16=$m timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_m16_${m}_$r.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/sam_m16_${m}_$r.json").read().strip().splitlines()[-1])
print("round $r M16=$m:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0], t["us_per_launch"], t["frac"]) for t in d["top_launches"] if "gemm256x192" in t["kernel"]])
PY
done; done
