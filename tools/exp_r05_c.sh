#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_mailbox_gpu.py tests/test_models_gpu.py -m gpu -q -x -k "soak_gan or bf16_mfma" 2>&1 | tail -80 > gpurun_out/r05_fail_c.log
cat gpurun_out/r05_fail_c.log
