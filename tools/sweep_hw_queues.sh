#!/bin/bash
# development aid: does the headline depend on the number of HIP hardware queues / launches in flight?
F="--no-cpu-baseline --no-train --no-end-to-end --no-files --steps 12 --warmup 4"
for q in 4 8; do for fl in 4 6; do
  echo "GPU_MAX_HW_QUEUES=$q inflight=$fl: $(GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py $F --inflight $fl 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "samples/s; per-sample ms in flight", round(d["roofline"]["serial_depth"]["per_sample_device_ms_with_4_launches_in_flight"],1))')"
done; done
