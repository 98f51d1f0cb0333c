#!/bin/bash
# steps per launch of the persistent kernel: 4 (one sample per workgroup) against 8 / 12 (two / three samples per workgroup, the tail shared)
run() { python bench.py "$@" --no-train --no-files --no-long --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1:], d['value'], d['ms_per_step'], d['slot_cycle']['kernel_ms'])" "$@"; }
for g in 4 8; do run --group $g --steps 20 --warmup 5; done
for g in 4 8 12 16; do run --group $g --steps 96 --warmup 16; done
run --group 8 --inflight 3 --steps 96 --warmup 24
