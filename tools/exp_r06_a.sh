#!/bin/bash
# round 6, first GPU call: CU-mask mapping, HW_ID placement, the CU-partitioned pipeline at several splits, the shipped bench for comparison
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06a
./tools/micro/cumask_probe > gpurun_out/r06a/cumask_probe.log 2>&1
./tools/micro/probe_hwid > gpurun_out/r06a/probe_hwid.log 2>&1
for rc in 0 24 32 40; do for nf in 2 3; do
  timeout 300 python tools/exp_cumask_pipeline.py --raster-cus $rc --policy stride --inflight $nf --steps 24 >> gpurun_out/r06a/pipeline.log 2>gpurun_out/r06a/pipeline_err_${rc}_${nf}.log
done; done
timeout 300 python tools/exp_cumask_pipeline.py --raster-cus 32 --policy tail --inflight 3 --steps 24 >> gpurun_out/r06a/pipeline.log 2>gpurun_out/r06a/pipeline_err_tail.log
timeout 600 python bench.py --no-train --no-files --no-pmc --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r06a/bench_short.json 2> gpurun_out/r06a/bench_short.err
cat gpurun_out/r06a/pipeline.log
