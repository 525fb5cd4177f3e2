#!/bin/bash
TAG=${1:-r3b}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q -rA > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 400 python bench.py --workload pipeline --no-cpu-baseline > $O/bench_pipeline.json 2> $O/bench_pipeline.err; echo "pipeline rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_pipeline.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"]); print(json.dumps(d["host_inclusive"], indent=1))
PY
