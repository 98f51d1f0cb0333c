#!/bin/bash
cd "$(dirname "$0")/.."
for b in 1 0; do
  echo "### OCTA_E2E_BURST=$b seg gen 512"; OCTA_E2E_BURST=$b python train_synthetic.py --steps 256 --warmup 128 --gen-batch 512 2>/dev/null | tail -1 | cut -c1-330
done
echo "### burst=0 gen 256 (round-4 shape)"; OCTA_E2E_BURST=0 python train_synthetic.py --steps 192 --warmup 64 --gen-batch 256 2>/dev/null | tail -1 | cut -c1-330
echo "### burst=1 gan gen 512"; OCTA_E2E_BURST=1 python train_synthetic.py --steps 128 --warmup 64 --gen-batch 512 --gan 2>/dev/null | tail -1 | cut -c1-330
echo "### burst=0 gan gen 128"; OCTA_E2E_BURST=0 python train_synthetic.py --steps 64 --warmup 32 --gen-batch 128 --gan 2>/dev/null | tail -1 | cut -c1-330
