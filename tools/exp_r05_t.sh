#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_models_gpu.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
OCTA_SKIP_TORCH=1 python tools/time_conv.py 4 2>&1 | grep -E "wgrad|mfma"
python tools/time_train.py 4 2>&1 | tail -1
python tools/time_train.py 4 2>&1 | tail -1
