#!/bin/bash
# Round-3 first GPU call: the parity suite, the bench lines touched this round, the SAM 2.1 HBM-traffic PMC passes.
TAG=${1:-r3a}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -rA > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 400 python bench.py > $O/bench_circuit.json 2> $O/bench_circuit.err; echo "circuit rc=$?"
timeout -k 10 400 python bench.py --workload pipeline --no-cpu-baseline > $O/bench_pipeline.json 2> $O/bench_pipeline.err; echo "pipeline rc=$?"
timeout -k 10 300 python bench.py --workload sam2l_box --dtype bf16 --no-cpu-baseline > $O/bench_sam2l_box_bf16.json 2> $O/bench_sam2l_box_bf16.err; echo "box bf16 rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/sam_fetch -o f -- python3 tools/one_step.py sam2l 2 > $O/sam_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/sam_write -o w -- python3 tools/one_step.py sam2l 2 > $O/sam_write.log 2>&1; echo "pmc write rc=$?"
python3 tools/traffic_sam.py $O/sam_fetch $O/sam_write 2 $O/sam_traffic.json "SAM 2.1 Hiera-L B=16 fp16, 2 eager passes, $TAG" > $O/sam_traffic.txt 2>&1; head -12 $O/sam_traffic.txt
rm -rf $O/sam_fetch/*/*.db $O/sam_write/*/*.db 2>/dev/null
du -sh $O
