#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_models_gpu.py -m gpu -q -k "never_leave or side_stream" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
bash tools/profile_bench.sh r05 > gpurun_out/r05_profile_bench.out 2>&1
bash tools/profile_train.sh r05 > gpurun_out/r05_profile_train.out 2>&1
bash tools/profile_gan.sh r05 > gpurun_out/r05_profile_gan.out 2>&1
tail -3 gpurun_out/r05_profile_bench.out | cut -c1-300
ls gpurun_out | grep r05_ | head -50
