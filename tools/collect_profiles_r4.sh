#!/bin/bash
# Copy the summaries of a tools/final_run_r4.sh result (gpurun_out/<tag>/) into profiles/ under the round's prefix.
# usage: tools/collect_profiles_r3.sh <tag> <prefix, e.g. r04>
TAG=$1; P=$2; O=gpurun_out/$TAG
for w in circuit yolo11n yolo11l sam2l sam2l_bf16 sam2l_box sam2l_box_bf16 pipeline; do
  [ -s $O/bench_$w.json ] && tail -1 $O/bench_$w.json > profiles/${P}_bench_$w.json
done
line() { python3 -c "import json,sys; d=json.loads(open('$1').read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')" 2>/dev/null; }
python3 tools/prof_summary.py $O/prof_yolo profiles/${P}_yolo11n_b32_kernel_stats.md "Command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload yolo11n --steps 20 --warmup 5 --no-cpu-baseline (YOLO11-n, B=32, fp16; graph replays + eager profiling passes).  Kernel names demangled (tools/kname.py).  Same run's bench line: $(line $O/prof_yolo.log)"
python3 tools/prof_summary.py $O/prof_sam profiles/${P}_sam2l_b16_kernel_stats.md "Command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload sam2l --steps 3 --warmup 1 --no-cpu-baseline (SAM 2.1 Hiera-L, B=16, fp16 operands / f32 residual stream).  Kernel names demangled (tools/kname.py): the spelling bench.py's roofline objects use.  Same run's bench line: $(line $O/prof_sam.log)"
python3 - <<PY
import json
t = json.load(open("$O/traffic.json"))
t["collected"] = "round 4 kernels ($TAG), YOLO11-n B=32 fp16, 3 eager steps under rocprofv3 --pmc"
json.dump(t, open("profiles/${P}_yolo11n_b32_traffic.json", "w"), indent=1)
json.dump(t, open("profiles/traffic_latest.json", "w"), indent=1)
s = json.load(open("$O/sam_traffic.json"))
json.dump(s, open("profiles/${P}_sam2l_b16_traffic.json", "w"), indent=1)
json.dump(s, open("profiles/sam_traffic_latest.json", "w"), indent=1)
print("yolo traffic", t["hbm_bytes_per_step"], "sam traffic per pass", s["hbm_bytes_per_pass"])
PY
cp $O/sam_traffic.txt profiles/${P}_sam2l_b16_traffic.txt
cp $O/batch_compare_sam2l.txt profiles/${P}_batch_compare_sam2l_8_vs_16.txt
cp $O/yolo_floor.txt profiles/${P}_yolo11n_floor.txt
tail -3 $O/pytest.log > profiles/${P}_pytest_gpu_tail.txt
python3 tools/pmc_table.py $O profiles/${P}_sam2l_pmc.md
ls -la profiles | grep ${P}_
