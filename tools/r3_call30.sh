#!/bin/bash
# the round-end driver's flow on the final code: GPU suite, smoke, default bench line
TAG=${1:-r3end}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); r=d['roofline']; print(d['value'], d['ms_per_step'], r['bound'], r['frac'], r['mfma_frac'], r['traffic'], r['us_per_launch'], d['cpu_baseline']['value'])"
