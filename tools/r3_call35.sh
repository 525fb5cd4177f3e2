#!/bin/bash
# tok_linear16 wave priorities: shipped (none) vs static prio 1 for waves 4-7 (p1) vs prio 1 in every epilogue phase (p2)
TAG=${1:-r3pr}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
L=circuitvision_amd/libcvmi355.so
cp $L /tmp/lib_ship.so
for v in ship p3 p4 ship p3 p4; do
  if [ $v = ship ]; then cp /tmp/lib_ship.so $L; else cp circuitvision_amd/libcvmi355_$v.so $L; fi
  timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_$v.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/sam_$v.json").read().strip().splitlines()[-1])
print("$v:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][20:52], t["us_per_launch"]) for t in d["top_launches"] if "tok_linear16" in t["kernel"]])
PY
done
cp /tmp/lib_ship.so $L
