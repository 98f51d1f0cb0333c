#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
echo "bench rc $? wall $(( $(date +%s) - T0 )) s; stdout lines: $(wc -l < gpurun_out/r05_bench_line.json)"
tail -3 gpurun_out/r05_bench_line.err
