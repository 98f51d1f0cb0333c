# GPU: the regular build's phase times at full occupancy, the 2 x 512 oracle digests, its L2-miss traffic, a short repro run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
TAG=${1:-cur}
timeout 300 python tools/sim_phases.py 512 2 2>&1 | grep -v amdgpu.ids | tail -n 3 | tee gpurun_out/r4/phases512_$TAG.log
timeout 600 python tools/validate_many.py --digests tools/cache/oracle_digests_1000_512.npz 2>&1 | grep RESULT | tee gpurun_out/r4/validate_$TAG.log
timeout 600 python tools/validate_many.py --digests tools/cache/oracle_digests_200000_512.npz 2>&1 | grep RESULT | tee -a gpurun_out/r4/validate_$TAG.log
timeout 600 python tools/sim_traffic.py default 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/traffic_$TAG.log
timeout 600 python tools/repro_sim_race.py ${2:-20} 2>&1 | grep -v amdgpu.ids | cut -c1-250 | tail -n 3 | tee gpurun_out/r4/repro_$TAG.log
