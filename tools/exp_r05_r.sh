#!/bin/bash
# wgrad sensitivity to the DMA's address pattern (timing only)
cd "$(dirname "$0")/.."
export OCTA_SKIP_TORCH=1
for v in ${VARS:-base wexp8}; do
  echo "== $v"
  if [ $v = base ]; then python tools/time_conv.py 4; else OCTA_HIP_LIB=$PWD/gpurun_variants/liboctahip_$v.so python tools/time_conv.py 4; fi 2>&1 | grep wgrad | awk '{print $7, $8}' | paste -sd' '
done
