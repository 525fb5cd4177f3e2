#!/bin/bash
# Copy the summaries of a tools/final_run.sh result (gpurun_out/<tag>/) into profiles/ under the round's prefix.
# usage: tools/collect_profiles.sh <tag> <prefix, e.g. r02>
TAG=$1; P=$2; O=gpurun_out/$TAG
for w in circuit yolo11n yolo11l sam2l sam2l_bf16 sam2l_box pipeline; do
  [ -s $O/bench_$w.json ] && tail -1 $O/bench_$w.json > profiles/${P}_bench_$w.json
done
python3 tools/prof_summary.py $O/prof_yolo profiles/${P}_yolo11n_b32_kernel_stats.md "Command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload yolo11n --steps 20 --warmup 5 --no-cpu-baseline (YOLO11-n, B=32, fp16; graph replays + eager profiling passes). Same run's bench line: $(python3 -c "import json,sys; d=json.loads(open('$O/prof_yolo.log').read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')" 2>/dev/null)"
python3 tools/prof_summary.py $O/prof_sam profiles/${P}_sam2l_b16_kernel_stats.md "Command: rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --workload sam2l --steps 3 --warmup 1 --no-cpu-baseline (SAM 2.1 Hiera-L, B=16, fp16 operands / f32 residual stream). Same run's bench line: $(python3 -c "import json,sys; d=json.loads(open('$O/prof_sam.log').read().strip().splitlines()[-1]); print(d['value'], d['unit'], d['ms_per_step'], 'ms/step')" 2>/dev/null)"
python3 - <<PY
import json, datetime
t = json.load(open("$O/traffic.json"))
t["collected"] = "round 2 final kernels ($TAG), YOLO11-n B=32 fp16, 3 eager steps under rocprofv3 --pmc"
json.dump(t, open("profiles/${P}_yolo11n_b32_traffic.json", "w"), indent=1)
json.dump(t, open("profiles/traffic_latest.json", "w"), indent=1)
print("traffic", t["hbm_bytes_per_step"])
PY
python3 tools/pmc_table.py $O profiles/${P}_sam2l_pmc.md
ls -la profiles | grep ${P}_
