#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( python tools/validate_many.py --digests tools/cache/oracle_digests_1000_512.npz --reps 2; echo "exit $?"; python tools/validate_many.py --digests tools/cache/oracle_digests_200000_512.npz --reps 2; echo "exit $?" ) 2>&1 | grep -E "RESULT|exit|differs|error" > gpurun_out/r05_validate.log
python tools/repro_sim_race.py 200 2>&1 | tail -8 > gpurun_out/r05_repro_sim_race.log
cat gpurun_out/r05_validate.log gpurun_out/r05_repro_sim_race.log
