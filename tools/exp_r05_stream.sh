#!/bin/bash
# several persistent kernels at a time (no gate), with the rasterisation planned ahead; exclusive and co-resident render workgroups
run() { python bench.py "$@" --steps 36 --warmup 12 --no-train --no-files --no-long --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['launches_in_flight'], d['config']['persistent_kernels_at_a_time'], d['value'], d['ms_per_step'], d['slot_cycle']['kernel_ms'])"; }
export OCTA_BENCH_PLAN_AHEAD=1
for lib in "" gpurun_variants/liboctahip_dense2.so; do
  export OCTA_HIP_LIB=$lib; [ -z "$lib" ] && unset OCTA_HIP_LIB
  echo "lib=$lib"
  run --inflight 3
  run --inflight 2 --no-serial-sim
  run --inflight 3 --no-serial-sim
  run --inflight 4 --no-serial-sim
done
