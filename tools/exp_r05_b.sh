#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -60 > gpurun_out/r05_suite_b.log
timeout 600 python -m pytest tests/test_models_gpu.py tests/test_networks_golden.py tests/test_fullsize_gpu.py -m gpu -q -s -k "same_weights or wrong_tap or mfma_networks or whole_network" 2>&1 | grep -E "^\[|logits:|passed|failed|Error|assert" > gpurun_out/r05_parity_numbers.log
cat gpurun_out/r05_suite_b.log; cat gpurun_out/r05_parity_numbers.log
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_b.json 2> gpurun_out/r05_bench_b.err; echo "bench rc $?"; tail -5 gpurun_out/r05_bench_b.err; cut -c1-600 gpurun_out/r05_bench_b.json
