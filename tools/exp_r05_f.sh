#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_raster_gpu.py tests/test_cli_gpu.py tests/test_training_cli_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
bash tools/raster_pmc.sh r05 | grep -E "render|TOTAL|ampl"
python tools/time_raster.py 2>&1 | grep -E "raster|dither" | tail -8
