#!/bin/bash
# edges per fold batch / cell-list capacity of the render kernel (variant libraries from tools/build_variant.py)
cd "$(dirname "$0")/.."
for v in base fb5_200 fb6_224 fb6_160; do
  echo "== $v"
  if [ $v = base ]; then python tools/time_raster.py; else OCTA_HIP_LIB=$PWD/gpurun_variants/liboctahip_$v.so python tools/time_raster.py; fi 2>&1 | grep -E "raster \[" | tail -3 | awk '{print $2,$3,$5,$6}' | paste -sd'|'
done
