cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 600 python tools/repro_sim_race.py 100 2>&1 | tail -n 2
timeout 600 python tools/validate_many.py --digests tools/cache/oracle_digests_1000_512.npz --reps 2 2>&1 | grep RESULT
timeout 900 python -m pytest tests/test_sim_gpu.py -m gpu -x -q 2>&1 | tail -n 2
