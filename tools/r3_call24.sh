#!/bin/bash
# qkv row pitch padded to whole 128-byte lines: parity (every SAM + pipeline test), then A/B on the segmenter bench
TAG=${1:-r3p}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_sam2_gpu.py tests/test_pipeline_gpu.py tests/test_boundary_gpu.py -x -q -m gpu > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for pd in 1 0 1 0; do
  CVMI_SAM_QKV_PAD=$pd timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_pd$pd.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/sam_pd$pd.json").read().strip().splitlines()[-1])
print("QKV_PAD=$pd:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][:60], t["launches_per_pass"], t["us_per_launch"]) for t in d["top_launches"] if "tok_linear_kernel" in t["kernel"] or "win16" in t["kernel"] or "res64" in t["kernel"]], d["stages"]["sam2l"]["breakdown"]["attn_window"]["ms"], d["stages"]["sam2l"]["breakdown"]["gemm"]["ms"])
PY
done
