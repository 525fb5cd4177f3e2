#!/bin/bash
# tok_linear16 weight-ring depth: 6 (shipped) vs 8 ds_read_b128 in flight -- two builds of the library, swapped on the box's scratch copy
TAG=${1:-r3pf}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
L=circuitvision_amd/libcvmi355.so
cp $L /tmp/lib_pf6.so
for v in pf6 pf8 pf6 pf8; do
  if [ $v = pf8 ]; then cp circuitvision_amd/libcvmi355_pf8.so $L; else cp /tmp/lib_pf6.so $L; fi
  timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_$v.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/sam_$v.json").read().strip().splitlines()[-1])
print("$v:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][20:52], t["us_per_launch"]) for t in d["top_launches"] if "tok_linear16" in t["kernel"]])
PY
done
cp /tmp/lib_pf6.so $L
