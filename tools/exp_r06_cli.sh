#!/bin/bash
# round 6: the CLI's on-disk rate with the launches ordered by pipeline.SimGate (device-side) against the slot gate, alternating; 8192 samples each
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
timeout 600 python -m pytest tests/test_cli_gpu.py -x -q 2>&1 | tail -2
for rep in 1 2; do for sg in 1 0; do
OCTA_CLI_SIM_GATE=$sg python - <<'PY' 2>> gpurun_out/r06/cli_gate.err | tee -a gpurun_out/r06/cli_gate.log
import sys, time, os, shutil, contextlib, io
sys.path.insert(0, os.getcwd())
import generate_vessel_graph
n = 8192
for rep in range(2):
    shutil.rmtree("/dev/shm/octa_cli_out", ignore_errors=True)
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        generate_vessel_graph.main(["--config_file", "docker/vessel_graph_gen_docker_config.yml", "--num_samples", str(n), "--labels", "--seed", "7000000", "--output.directory", "/dev/shm/octa_cli_out"])
    dt = time.time() - t
    print(f"sim_gate {os.environ['OCTA_CLI_SIM_GATE']} rep {rep}: {len(os.listdir('/dev/shm/octa_cli_out'))} dirs, {n / dt:.1f} triples/s", flush=True)
shutil.rmtree("/dev/shm/octa_cli_out", ignore_errors=True)
PY
done; done
