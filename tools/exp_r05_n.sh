#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^Initialized\|^Loaded\|amdgpu.ids" | tail -6 > gpurun_out/r05_gpu_suite.log
cat gpurun_out/r05_gpu_suite.log
