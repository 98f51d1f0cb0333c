#!/bin/bash
# round 6: sub-phase profiles of the simulator at full occupancy (512 samples); usage: tools/exp_r06_c.sh variant...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06c
for v in "$@"; do
  echo "== $v" >> gpurun_out/r06c/phases_$v.log
  OCTA_HIP_LIB=$PWD/gpurun_variants/liboctahip_$v.so OCTA_PHASES_RAW=1 timeout 300 python tools/sim_phases.py 512 1 2>&1 | grep -v "amdgpu.ids\|^\[octa\]" >> gpurun_out/r06c/phases_$v.log
  cat gpurun_out/r06c/phases_$v.log
done
