#!/bin/bash
TAG=${1:-r3k}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
for r in 1 2; do for ns in 1 4 3; do
  timeout -k 10 300 python bench.py --streams $ns --no-cpu-baseline --no-profile-pass --steps 10 > $O/circuit_s${ns}_$r.json 2>$O/err_s${ns}_$r.txt
  python - <<PY
import json
try:
    d=json.loads(open("$O/circuit_s${ns}_$r.json").read().strip().splitlines()[-1])
    print("round $r streams=$ns:", d["value"], "images/s", d["ms_per_step"], "ms/step")
except Exception as e:
    print("round $r streams=$ns failed", e, open("$O/err_s${ns}_$r.txt").read()[-500:])
PY
done; done
