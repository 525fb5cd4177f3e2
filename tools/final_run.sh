set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fin
python bench.py > gpurun_out/fin/bench_yolo11n.json 2> gpurun_out/fin/bench_yolo11n.err
python bench.py --workload yolo11l --no-cpu-baseline > gpurun_out/fin/bench_yolo11l.json 2>/dev/null
python bench.py --workload sam2l --no-cpu-baseline > gpurun_out/fin/bench_sam2l.json 2>/dev/null
python bench.py --workload sam2l_box --no-cpu-baseline > gpurun_out/fin/bench_sam2l_box.json 2>/dev/null
echo benches done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin/prof_yolo -o y -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/fin/prof_yolo.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin/prof_sam -o s -- python3 bench.py --workload sam2l --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/fin/prof_sam.log 2>&1
echo profiles done
