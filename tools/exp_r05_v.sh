#!/bin/bash
# kernel-trace stats of the U-Net step (no counter passes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
rm -rf gpurun_out/ptrain_kt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ptrain_kt -- python tools/time_train.py 4 1216 > gpurun_out/r05_train_kt.log 2>&1
find gpurun_out/ptrain_kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05_train_kernel_stats.csv
rm -rf gpurun_out/ptrain_kt
tail -1 gpurun_out/r05_train_kt.log
