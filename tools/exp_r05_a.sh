#!/bin/bash
# round 5, call 1: side-stream weight gradients / two-stream GAN step A/B, then the GPU suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r05_exp_a.log
: > $L
run() { echo "### $*" >> $L; ( env "$@" ) >> $L 2>&1; }
run OCTA_WGRAD_STREAM=1 python tools/time_train.py 4
run OCTA_WGRAD_STREAM=0 python tools/time_train.py 4
run OCTA_WGRAD_STREAM=1 python tools/time_train.py 8
run OCTA_WGRAD_STREAM=0 python tools/time_train.py 8
run OCTA_WGRAD_STREAM=1 OCTA_GAN_STREAMS=1 python tools/time_gan.py 4
run OCTA_WGRAD_STREAM=0 OCTA_GAN_STREAMS=1 python tools/time_gan.py 4
run OCTA_WGRAD_STREAM=1 OCTA_GAN_STREAMS=0 python tools/time_gan.py 4
run OCTA_WGRAD_STREAM=0 OCTA_GAN_STREAMS=0 python tools/time_gan.py 4
cat $L
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r05_suite_a.log
