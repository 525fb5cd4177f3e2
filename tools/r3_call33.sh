#!/bin/bash
# tok_linear16 A/B: shipped library vs circuitvision_amd/libcvmi355_old.so (built from the previous source): parity, then interleaved bench runs
TAG=${1:-r3sw}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_sam2_gpu.py -x -q -m gpu -k "tok_linear or hiera or statistics or permutation or wrapper or golden" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
L=circuitvision_amd/libcvmi355.so
cp $L /tmp/lib_new.so
for v in new old new old; do
  if [ $v = new ]; then cp /tmp/lib_new.so $L; else cp circuitvision_amd/libcvmi355_old.so $L; fi
  timeout -k 10 300 python bench.py --workload sam2l --no-cpu-baseline --steps 5 > $O/sam_$v.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/sam_$v.json").read().strip().splitlines()[-1])
print("$v:", d["ms_per_step"], "ms/step;", [(t["kernel"].split(":")[0][20:52], t["us_per_launch"]) for t in d["top_launches"] if "tok_linear16" in t["kernel"]])
PY
done
cp /tmp/lib_new.so $L
