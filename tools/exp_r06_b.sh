#!/bin/bash
# round 6: is the un-gated pipeline (plain streams, N launches in flight, no lock / polling / sleeps) reproducible? 5 runs each, and the shipped gate 3 times
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06b
for rep in 1 2 3 4 5; do for nf in 2 3; do
  timeout 300 python tools/exp_cumask_pipeline.py --raster-cus 0 --inflight $nf --steps 20 --warmup 5 >> gpurun_out/r06b/nogate.log 2>>gpurun_out/r06b/nogate.err
done; done
for rep in 1 2 3; do
timeout 600 python bench.py --no-train --no-files --no-pmc --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('gated', d['value'], d['slot_cycle']['render_enqueue_ms'], d['cu_time']['simulator_share'])" >> gpurun_out/r06b/nogate.log
done
cat gpurun_out/r06b/nogate.log
