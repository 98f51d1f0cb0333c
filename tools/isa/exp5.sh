cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 600 python tools/sim_phases.py 512 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/phases512_b.log
timeout 900 python -m pytest tests/test_sim_gpu.py -m gpu -x -q -k "full_length or wide_reference or short_runs or deterministic or batch_vs_oracle" 2>&1 | tail -n 3
