cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 300 python tools/sim_phases.py 512 2 2>&1 | grep -v amdgpu.ids | tail -n 3 | tee gpurun_out/r4/phases512_c.log
timeout 600 python tools/validate_many.py --digests tools/cache/oracle_digests_1000_512.npz --reps 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/validate_c.log
timeout 900 python -m pytest tests/test_sim_gpu.py -m gpu -x -q 2>&1 | tail -n 3
