#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py tests/test_networks_golden.py tests/test_fallbacks.py -m gpu -q 2>&1 | grep -E "passed|failed|Error|^FAILED" | tail -12
OCTA_SKIP_TORCH=1 python tools/time_conv.py 4 2>&1 | grep -E "wgrad|mfma"
python tools/time_train.py 4 2>&1 | tail -1
python tools/time_train.py 4 2>&1 | tail -1
