#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r05_exp_e.log
: > $L
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q -x 2>&1 | tail -15 >> $L
run() { echo "### $*" >> $L; ( env "$@" ) 2>&1 | grep -E "DynUNet train|GAN-seg step|Error|error" >> $L; }
run OCTA_EPI_STATS=0 python tools/time_train.py 4
run OCTA_EPI_STATS=1 python tools/time_train.py 4
run OCTA_EPI_STATS=0 python tools/time_train.py 4
run OCTA_EPI_STATS=1 python tools/time_train.py 4
run OCTA_EPI_STATS=1 OCTA_STAT_SLOTS=4 python tools/time_train.py 4
run OCTA_EPI_STATS=1 OCTA_STAT_SLOTS=64 python tools/time_train.py 4
run OCTA_EPI_STATS=0 python tools/time_gan.py 4
run OCTA_EPI_STATS=1 python tools/time_gan.py 4
cat $L
