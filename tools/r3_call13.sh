#!/bin/bash
TAG=${1:-r3n}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for r in 1 2; do for ns in 1 2 3 4; do
  timeout -k 10 300 python bench.py --streams $ns --no-cpu-baseline --no-profile-pass --steps 10 > $O/c_${ns}_$r.json 2>$O/e_${ns}_$r.txt
  python3 - <<PY
import json
try:
    d=json.loads(open("$O/c_${ns}_$r.json").read().strip().splitlines()[-1]); print("round $r streams=$ns (detector graph linear for > 1):", d["value"], "images/s", d["ms_per_step"], "ms/step")
except Exception as e: print("failed $ns", open("$O/e_${ns}_$r.txt").read()[-600:])
PY
done; done
