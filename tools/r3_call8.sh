#!/bin/bash
TAG=${1:-r3h}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "gemm or row_statistics or tok_linear" > $O/pytest_a.log 2>&1; echo "pytest gemm rc=$?"; tail -4 $O/pytest_a.log
CVMI_TOKLIN_M16=0 timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "row_statistics" > $O/pytest_b.log 2>&1; echo "pytest stats (old consumer) rc=$?"; tail -2 $O/pytest_b.log
CVMI_G192_M16=0 timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "row_statistics" > $O/pytest_c.log 2>&1; echo "pytest stats (relic producer) rc=$?"; tail -2 $O/pytest_c.log
timeout -k 10 900 python -m pytest tests/test_sam2_gpu.py -m gpu -q -x -k "hiera_l or replay" > $O/pytest_sam.log 2>&1; echo "pytest sam rc=$?"; tail -3 $O/pytest_sam.log
for r in 1 2; do
  timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_$r.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/sam_$r.json").read().strip().splitlines()[-1])
print("round $r:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:40], t["us_per_launch"]) for t in d["top_launches"]])
PY
done
