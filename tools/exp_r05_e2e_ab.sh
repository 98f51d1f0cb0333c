#!/bin/bash
# on-the-fly loops of the current tree against the tree in gpurun_variants/old (a git worktree of an earlier commit, built in place), same box
run() { (cd "$1" && python -c "
import train_synthetic
r = train_synthetic.run(steps=$3, batch=4, gen_batch=512, seed0=500000, log=False, warmup=$4, gan=$2)
print('$1', 'gan=$2', round(r['value'], 1), 'imgs/s', round(r['ms_per_step'], 2), 'ms/step')
" 2>/dev/null | tail -1); }
for i in 1 2; do
  run . False 384 128
  run gpurun_variants/old False 384 128
done
for i in 1 2; do
  run . True 160 48
  run gpurun_variants/old True 160 48
done
