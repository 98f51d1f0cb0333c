#!/bin/bash
# development aid: does the headline depend on the number of HIP hardware queues / launches in flight?
F="--no-cpu-baseline --no-train --no-end-to-end --no-files --steps 16 --warmup 6"
for q in ${QUEUES:-4 8}; do for fl in ${INFLIGHT:-4 5 6 8}; do
  echo "GPU_MAX_HW_QUEUES=$q inflight=$fl: $(GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py $F --inflight $fl 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), "samples/s; simulator share of CU time", round(d["cu_time"]["simulator_share"],3), "span ms", round(d["cu_time"]["span_ms_per_sample"],1))')"
done; done
