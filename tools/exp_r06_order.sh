#!/bin/bash
# round 6: the device-side ordering of rasterisation behind the next launch: its tests, then the driver's headline leg N times
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_pipeline_gpu.py -x -q 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/r06/order_tests.log
for rep in $(seq 1 ${1:-5}); do
timeout 600 python bench.py --no-train --no-files --no-pmc --no-cpu-baseline --steps 20 --warmup 5 2>gpurun_out/r06/order_bench_$rep.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('run', d['value'], 'kernel', d['roofline']['avg_launch_ms'], d['slot_cycle']['render_enqueue_ms'], d['slot_cycle']['render_wait_ms'], d['cu_time']['simulator_share'])" | tee -a gpurun_out/r06/order_runs.log
done
