cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
./tools/micro/probe_hwid > gpurun_out/r4/probe_hwid.log 2>&1
cat gpurun_out/r4/probe_hwid.log
timeout 600 python tools/sim_phases.py 512 2 > gpurun_out/r4/phases512.log 2>&1
cat gpurun_out/r4/phases512.log
timeout 600 python tools/sim_phases.py 128 2 > gpurun_out/r4/phases128.log 2>&1
cat gpurun_out/r4/phases128.log
