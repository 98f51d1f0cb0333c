#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_models_gpu.py tests/test_mailbox_gpu.py -m gpu -q -k "gan or soak" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for i in 1 2; do
OCTA_GAN_STREAMS=1 python tools/time_gan.py 4 2>&1 | grep "GAN-seg step"
OCTA_GAN_STREAMS=0 python tools/time_gan.py 4 2>&1 | grep "GAN-seg step"
done
