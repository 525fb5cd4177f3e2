#!/bin/bash
TAG=${1:-r3d}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_sam2_gpu.py -m gpu -q -rA -k "fp8 or attention" > $O/pytest_fp8.log 2>&1; echo "pytest fp8 rc=$?"; grep -E "fp8 AV|fp8 attention|passed|failed|Error" $O/pytest_fp8.log | head -30
for r in 1 2; do for at in 16 fp8; do for dt in f16 bf16; do
  timeout -k 10 300 python bench.py --workload sam2l --dtype $dt --attn $at --no-cpu-baseline --steps 5 > $O/sam_${dt}_${at}_$r.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/sam_${dt}_${at}_$r.json").read().strip().splitlines()[-1])
print("round $r dtype $dt attn $at:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0], t["us_per_launch"]) for t in d["top_launches"] if "attn" in t["kernel"]], d["rooflines"]["sam2l_attention_global"]["kernel_ms_per_step"], d["rooflines"]["sam2l_attention_window"]["kernel_ms_per_step"])
PY
done; done; done
timeout -k 10 300 python bench.py --workload sam2l_box --dtype bf16 --attn fp8 --no-cpu-baseline > $O/bench_sam2l_box_bf16_fp8.json 2>/dev/null; echo "box bf16 fp8 rc=$?"; python -c "
import json; d=json.loads(open('$O/bench_sam2l_box_bf16_fp8.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
