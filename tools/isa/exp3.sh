cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 900 python tools/repro_sim_race.py 400 > gpurun_out/r4/repro_final.log 2>&1
tail -n 2 gpurun_out/r4/repro_final.log
timeout 1500 python -m pytest tests/test_sim_gpu.py tests/test_mailbox_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > gpurun_out/r4/sim_tests.log 2>&1
tail -n 5 gpurun_out/r4/sim_tests.log
timeout 900 python bench.py > gpurun_out/r4/bench1.json 2> gpurun_out/r4/bench1.err
tail -c 3000 gpurun_out/r4/bench1.json
