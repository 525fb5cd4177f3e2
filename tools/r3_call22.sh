#!/bin/bash
# q pre-scaled by scale * log2 e (cvmi_attn_desc.q_log2): parity (ops + every SAM test), then A/B on the segmenter bench
TAG=${1:-r3q}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_sam2_gpu.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -3 $O/pytest.log
for ql in 1 0 1 0; do
  CVMI_SAM_QLOG2=$ql timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_ql$ql.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/sam_ql$ql.json").read().strip().splitlines()[-1])
print("QLOG2=$ql:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:40], t["launches_per_pass"], t["us_per_launch"], t["bound"], t["frac"]) for t in d["top_launches"] if "attn" in t["kernel"]])
PY
done
