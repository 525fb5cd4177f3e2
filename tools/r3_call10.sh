#!/bin/bash
TAG=${1:-r3j}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_boundary_gpu.py -m gpu -q -k "gemm_256 or row_statistics or rank_without" > $O/pytest_a.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_a.log
for r in 1 2 3; do for rv in 1 0; do
  CVMI_G192_REV=$rv timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_rev${rv}_$r.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/sam_rev${rv}_$r.json").read().strip().splitlines()[-1])
print("round $r REV=$rv:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:34], t["us_per_launch"]) for t in d["top_launches"][:5]])
PY
done; done
