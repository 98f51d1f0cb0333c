#!/bin/bash
# on-the-fly loops with the generator's persistent kernel limited to fewer workgroups (OCTA_SIM_GRID): one per CU leaves half of every CU's
# LDS and registers to the training kernels
run() { python -c "
import train_synthetic
r = train_synthetic.run(steps=$2, batch=4, gen_batch=${GEN:-512}, seed0=500000, log=False, warmup=$3, gan=$1)
print('grid', '${OCTA_SIM_GRID:-512}', 'gen', ${GEN:-512}, 'gan=$1', round(r['value'], 1), 'imgs/s', round(r['ms_per_step'], 2), 'ms/step')
" 2>/dev/null | tail -1; }
for g in 512 256 128; do export OCTA_SIM_GRID=$g; run False 384 128; done
for g in 512 256 128; do export OCTA_SIM_GRID=$g; run True 160 48; done
