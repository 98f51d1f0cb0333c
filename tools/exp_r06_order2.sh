#!/bin/bash
# round 6: device-side ordering with the rasterisation planned ahead (host wait before the gate kernel) against the plain form, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
for rep in 1 2 3; do for pa in 1 0; do
OCTA_BENCH_PLAN_AHEAD=$pa timeout 600 python bench.py --no-train --no-files --no-pmc --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('plan_ahead $pa', d['value'], 'kernel', d['roofline']['avg_launch_ms'], d['slot_cycle']['render_enqueue_ms'], d['slot_cycle']['render_wait_ms'], d['cu_time']['simulator_share'])" | tee -a gpurun_out/r06/order2_runs.log
done; done
