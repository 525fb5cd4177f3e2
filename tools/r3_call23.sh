#!/bin/bash
# de-phased first round (stagger_first_round) on the fused stage-1 / 2 MLP: parity, then A/B over the step
TAG=${1:-r3s1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "mlp or tok_linear or hiera" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for st in 0 -1 8 24 0 -1; do
  CVMI_STAGGER_MLP=$st timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_st$st.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/sam_st$st.json").read().strip().splitlines()[-1])
print("STAGGER_MLP=$st:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:40], t["launches_per_pass"], t["us_per_launch"], t["bound"], t["frac"]) for t in d["top_launches"] if "hiera_mlp" in t["kernel"]])
PY
done
