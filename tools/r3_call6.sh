#!/bin/bash
TAG=${1:-r3f}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_sam2_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for r in 1 2; do
  timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_$r.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/sam_$r.json").read().strip().splitlines()[-1])
print("round $r:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:40], t["us_per_launch"]) for t in d["top_launches"]])
PY
done
