#!/bin/bash
# round 6: the regular build after a simulator change: phase times at full occupancy, 2 x 512 oracle digests, (optionally) the simulator's -m gpu tests and a repro run
# usage: tools/sim_check.sh TAG [tests] [repro_reps]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
TAG=${1:-cur}
OCTA_PHASES_RAW=1 timeout 300 python tools/sim_phases.py 512 2 2>&1 | grep -v "amdgpu.ids\|^\[octa\]" | tee gpurun_out/r06/phases512_$TAG.log
timeout 600 python tools/validate_many.py --digests tools/cache/oracle_digests_1000_512.npz 2>&1 | grep RESULT | tee gpurun_out/r06/validate_$TAG.log
timeout 600 python tools/validate_many.py --digests tools/cache/oracle_digests_200000_512.npz 2>&1 | grep RESULT | tee -a gpurun_out/r06/validate_$TAG.log
if [ "$2" = "tests" ]; then timeout 1500 python -m pytest tests/test_sim_gpu.py -x -q 2>&1 | tail -5 | tee gpurun_out/r06/tests_$TAG.log; fi
if [ -n "$3" ]; then timeout 900 python tools/repro_sim_race.py $3 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tail -n 3 | tee gpurun_out/r06/repro_$TAG.log; fi
