#!/bin/bash
TAG=${1:-r3g}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "tok_linear or row_statistics or pool" > $O/pytest_tl.log 2>&1; echo "pytest tok_linear rc=$?"; tail -4 $O/pytest_tl.log
timeout -k 10 900 python -m pytest tests/test_sam2_gpu.py -m gpu -q -x -k "hiera_l or replay" > $O/pytest_sam.log 2>&1; echo "pytest sam rc=$?"; tail -3 $O/pytest_sam.log
for r in 1 2; do for m in 1 0; do
  CVMI_TOKLIN_M16=$m timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_tl16_${m}_$r.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/sam_tl16_${m}_$r.json").read().strip().splitlines()[-1])
print("round $r TOKLIN_M16=$m:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:44], t["us_per_launch"]) for t in d["top_launches"] if "tok_linear" in t["kernel"]])
PY
done; done
