#!/bin/bash
# Shader clock and power while a workload runs: samples rocm-smi every 0.5 s beside `python bench.py <args>` (one gpurun call).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/clk
python bench.py "$@" --steps 300 --warmup 5 --no-cpu-baseline > gpurun_out/clk/bench.json 2> gpurun_out/clk/bench.err &
BP=$!
for i in $(seq 1 400); do
  kill -0 $BP 2>/dev/null || break
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo
  sleep 0.5
done > gpurun_out/clk/samples.txt
wait $BP
tail -c 300 gpurun_out/clk/bench.json | head -c 300; echo
grep -c . gpurun_out/clk/samples.txt; tail -25 gpurun_out/clk/samples.txt | cut -c1-200
