#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^Initialized\|^Loaded\|amdgpu.ids" | tail -15 > gpurun_out/r05_gpu_suite.log
cat gpurun_out/r05_gpu_suite.log
