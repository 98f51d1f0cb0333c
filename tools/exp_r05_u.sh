#!/bin/bash
cd "$(dirname "$0")/.."
export OCTA_SKIP_TORCH=1 OCTA_SKIP_WGRAD=1
for v in ${VARS:-base fexp1 fexp3}; do
  echo "== $v"
  if [ $v = base ]; then python tools/time_conv.py 4; else OCTA_HIP_LIB=$PWD/gpurun_variants/liboctahip_$v.so python tools/time_conv.py 4; fi 2>&1 | grep "mfma" | awk '{print $1,$2,$3,$6,$7,$8,$9}' | paste -sd'|'
done
