#!/bin/bash
# wgrad sensitivity: base / no shifted-B reads (wrong results, timing only) / no in-loop DMA / CW variants
cd "$(dirname "$0")/.."
export OCTA_SKIP_TORCH=1
for v in base wexp1 wexp3 cw1 cw2; do
  echo "== $v"
  case $v in
    base) python tools/time_conv.py 4 ;;
    cw1) OCTA_WGRAD_CW=1 python tools/time_conv.py 4 ;;
    cw2) OCTA_WGRAD_CW=2 python tools/time_conv.py 4 ;;
    *) OCTA_HIP_LIB=$PWD/gpurun_variants/liboctahip_$v.so python tools/time_conv.py 4 ;;
  esac 2>&1 | grep wgrad | awk '{print $5, $6, $7, $8, $9}' | paste -sd' '
done
