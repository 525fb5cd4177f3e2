#!/bin/bash
# Round-end measurement set (one gpurun call): bench lines of every workload, rocprofv3 kernel stats of the default workload's two
# stages, the YOLO11-n HBM-traffic PMC passes, SQ counters of the SAM pass.  Everything lands under gpurun_out/<tag>/; the summaries to
# be judged are copied into profiles/ afterwards (tools/prof_summary.py, tools/traffic.py, tools/pmc_table.py).
TAG=${1:-fin}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
python bench.py > $O/bench_circuit.json 2> $O/bench_circuit.err; echo "circuit rc=$?"
python bench.py --workload yolo11n --no-cpu-baseline > $O/bench_yolo11n.json 2>/dev/null; echo "yolo11n rc=$?"
python bench.py --workload yolo11l --no-cpu-baseline > $O/bench_yolo11l.json 2>/dev/null; echo "yolo11l rc=$?"
python bench.py --workload sam2l --no-cpu-baseline > $O/bench_sam2l.json 2>/dev/null; echo "sam2l rc=$?"
python bench.py --workload sam2l --dtype bf16 --no-cpu-baseline > $O/bench_sam2l_bf16.json 2>$O/bench_sam2l_bf16.err; echo "sam2l bf16 rc=$?"
python bench.py --workload sam2l_box --no-cpu-baseline > $O/bench_sam2l_box.json 2>/dev/null; echo "sam2l_box rc=$?"
python bench.py --workload pipeline --no-cpu-baseline > $O/bench_pipeline.json 2>/dev/null; echo "pipeline rc=$?"
echo benches done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_yolo -o y -- python3 bench.py --workload yolo11n --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_yolo.log 2>&1; echo "prof yolo rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sam -o s -- python3 bench.py --workload sam2l --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_sam.log 2>&1; echo "prof sam rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 tools/one_step.py yolo11n 3 > $O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 tools/one_step.py yolo11n 3 > $O/pmc_write.log 2>&1; echo "pmc write rc=$?"
python3 tools/traffic.py $O/pmc_fetch $O/pmc_write 3 $O/traffic.json > $O/traffic.log 2>&1; cat $O/traffic.log | tail -1
bash tools/pmc_sam.sh $TAG > $O/pmc_sam.log 2>&1; echo "pmc sam rc=$?"
ls $O
