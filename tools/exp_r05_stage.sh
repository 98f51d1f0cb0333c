#!/bin/bash
# the run's host part staged ahead (octa_sim_stage) on / off, with the per-workgroup share of the batch in the persistent kernel
run() { python bench.py "$@" --no-train --no-files --no-long --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json,os; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stage', os.environ.get('OCTA_SIM_STAGE'), sys.argv[1:], d['value'], d['ms_per_step'], d['slot_cycle']['kernel_ms'], d['slot_cycle']['sim_call_ms'])" "$@"; }
for i in 1 2 3; do for st in 0 1; do OCTA_SIM_STAGE=$st run --steps 20 --warmup 5; done; done
for st in 0 1; do OCTA_SIM_STAGE=$st run --steps 96 --warmup 16; OCTA_SIM_STAGE=$st run --inflight 3 --steps 96 --warmup 24; done
