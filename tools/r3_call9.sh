#!/bin/bash
TAG=${1:-r3i}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "attention or row_statistics" > $O/pytest_a.log 2>&1; echo "pytest attention rc=$?"; tail -3 $O/pytest_a.log
CVMI_G192_M16=0 timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "row_statistics" > $O/pytest_c.log 2>&1; echo "pytest stats (relic producer) rc=$?"; tail -2 $O/pytest_c.log
timeout -k 10 900 python -m pytest tests/test_sam2_gpu.py -m gpu -q -k "hiera or replay or mini" > $O/pytest_sam.log 2>&1; echo "pytest sam rc=$?"; tail -3 $O/pytest_sam.log
for r in 1 2; do for df in 8 0; do
  CVMI_ATTN_DEFER=$df timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_defer${df}_$r.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/sam_defer${df}_$r.json").read().strip().splitlines()[-1])
print("round $r DEFER=$df:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:40], t["us_per_launch"]) for t in d["top_launches"] if "attn" in t["kernel"]], d["rooflines"]["sam2l_attention_window"]["kernel_ms_per_step"], d["rooflines"]["sam2l_attention_global"]["kernel_ms_per_step"])
PY
done; done
