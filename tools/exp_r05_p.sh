#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
bash tools/raster_pmc.sh r05 > /dev/null 2>&1
T0=$(date +%s)
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
echo "bench rc $? wall $(( $(date +%s) - T0 )) s; stdout lines: $(wc -l < gpurun_out/r05_bench_line.json)"
bash tools/profile_bench.sh r05 > gpurun_out/r05_profile_bench.out 2>&1
bash tools/profile_train.sh r05 > gpurun_out/r05_profile_train.out 2>&1
bash tools/profile_gan.sh r05 > gpurun_out/r05_profile_gan.out 2>&1
cat gpurun_out/r05_raster_pmc.log
