#!/bin/bash
TAG=${1:-r3p}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_pipeline_gpu.py tests/test_boundary_gpu.py tests/test_yolo_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for ns in 3 1; do
timeout -k 10 400 python bench.py --streams $ns --workload pipeline --no-cpu-baseline --no-profile-pass > $O/bench_pipeline_$ns.json 2> $O/bench_pipeline_$ns.err; echo "pipeline streams=$ns rc=$?"
python3 - <<PY
import json
d=json.loads(open("$O/bench_pipeline_$ns.json").read().strip().splitlines()[-1])
print("graph-only", d["value"], d["ms_per_step"], "host-inclusive", d["host_inclusive"]["images_per_s"], d["host_inclusive"]["ratio_vs_graph_only"]); print(json.dumps(d["host_inclusive"]["phases_ms_per_step"]))
PY
done
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_circuit.json 2> $O/bench_circuit.err; echo "circuit rc=$?"; python3 -c "
import json; d=json.loads(open('$O/bench_circuit.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['streams'], d['roofline']['frac'], d['roofline']['traffic'])"
