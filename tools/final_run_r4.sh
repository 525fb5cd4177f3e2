#!/bin/bash
# Round-4 measurement set (one gpurun call): parity suite + smoke, bench lines of every workload, rocprofv3 kernel stats of the two stages,
# HBM-traffic PMC passes (YOLO11-n per step, SAM 2.1-L per launch), SQ counters of the SAM pass.  Everything lands under gpurun_out/<tag>/;
# tools/collect_profiles_r4.sh copies the summaries to be judged into profiles/.
TAG=${1:-fin4}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_circuit.json 2> $O/bench_circuit.err; echo "circuit rc=$?"
for w in yolo11n yolo11l sam2l sam2l_box; do
  timeout -k 10 400 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2>/dev/null; echo "$w rc=$?"
done
timeout -k 10 500 python bench.py --workload pipeline --total-images 64 --scaling-proxy 8 --no-cpu-baseline > $O/bench_pipeline.json 2> $O/bench_pipeline.err; echo "pipeline (+ 8-rank share) rc=$?"
timeout -k 10 300 python tools/batch_compare.py --workload sam2l --batches 8 16 > $O/batch_compare_sam2l.txt 2>&1; echo "batch compare rc=$?"
timeout -k 10 300 python tools/yolo_floor.py > $O/yolo_floor.txt 2>&1; echo "yolo floor rc=$?"
timeout -k 10 300 python bench.py --workload sam2l --dtype bf16 --no-cpu-baseline > $O/bench_sam2l_bf16.json 2>/dev/null; echo "sam2l bf16 rc=$?"
timeout -k 10 300 python bench.py --workload sam2l_box --dtype bf16 --no-cpu-baseline > $O/bench_sam2l_box_bf16.json 2>/dev/null; echo "sam2l_box bf16 rc=$?"
echo benches done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_yolo -o y -- python3 bench.py --workload yolo11n --steps 20 --warmup 5 --no-cpu-baseline > $O/prof_yolo.log 2>&1; echo "prof yolo rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sam -o s -- python3 bench.py --workload sam2l --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_sam.log 2>&1; echo "prof sam rc=$?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 tools/one_step.py yolo11n 3 > $O/pmc_fetch.log 2>&1; echo "pmc fetch rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 tools/one_step.py yolo11n 3 > $O/pmc_write.log 2>&1; echo "pmc write rc=$?"
python3 tools/traffic.py $O/pmc_fetch $O/pmc_write 3 $O/traffic.json > $O/traffic.log 2>&1; tail -1 $O/traffic.log
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/sam_fetch -o f -- python3 tools/one_step.py sam2l 2 > $O/sam_fetch.log 2>&1; echo "sam pmc fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/sam_write -o w -- python3 tools/one_step.py sam2l 2 > $O/sam_write.log 2>&1; echo "sam pmc write rc=$?"
python3 tools/traffic_sam.py $O/sam_fetch $O/sam_write 2 $O/sam_traffic.json "SAM 2.1 Hiera-L B=16 fp16, 2 eager passes, $TAG" > $O/sam_traffic.txt 2>&1; head -8 $O/sam_traffic.txt
bash tools/pmc_sam.sh $TAG > $O/pmc_sam.log 2>&1; echo "pmc sam rc=$?"
find $O -name "*.db" -delete 2>/dev/null
du -sh $O
