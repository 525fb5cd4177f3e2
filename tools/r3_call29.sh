#!/bin/bash
# fc2 with three A buffers (gemm256x192r3_kernel): parity, then A/B against the two-stage persistent form
TAG=${1:-r3g3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_sam2_gpu.py -x -q -m gpu -k "row_statistics or permutation or gemm or conv or hiera_l" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for dp in 1 0 1 0; do
  CVMI_G192_DEEP=$dp timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_dp$dp.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/sam_dp$dp.json").read().strip().splitlines()[-1])
print("DEEP=$dp:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:30], t["us_per_launch"]) for t in d["top_launches"] if "gemm256x192r" in t["kernel"]])
PY
done
CVMI_G192_DIAG=2 timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 3 > $O/sam_diag2.json 2>/dev/null
python3 - <<PY
import json
d=json.loads(open("$O/sam_diag2.json").read().strip().splitlines()[-1])
print("DEEP=1 DIAG=2:", [(t["kernel"].split(":")[0][:30], t["us_per_launch"]) for t in d["top_launches"] if "gemm256x192r" in t["kernel"]])
PY
