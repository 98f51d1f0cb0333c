#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "^Initialized\|^Loaded\|amdgpu.ids" > gpurun_out/r05_suite_d_full.log
tail -30 gpurun_out/r05_suite_d_full.log
grep -E "^\[mfma golden\]|^\[generator|^\[discriminator|^logits:" gpurun_out/r05_suite_d_full.log
