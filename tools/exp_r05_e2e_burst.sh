#!/bin/bash
# on-the-fly loops: the trainer steps aside while the generator's persistent kernel runs (OCTA_E2E_BURST=1) against sharing the GPU
run() { python -c "
import train_synthetic
r = train_synthetic.run(steps=$2, batch=4, gen_batch=512, seed0=500000, log=False, warmup=$3, gan=$1)
print('burst', '${OCTA_E2E_BURST:-0}', 'gan=$1', round(r['value'], 1), 'imgs/s', round(r['ms_per_step'], 2), 'ms/step')
" 2>/dev/null | tail -1; }
for i in 1 2; do for b in 0 1; do export OCTA_E2E_BURST=$b; run False 384 128; done; done
for i in 1 2; do for b in 0 1; do export OCTA_E2E_BURST=$b; run True 160 48; done; done
