#!/bin/bash
# round 6: the on-disk triples through the CLI with the native batch writer; writer-thread sweep; the headline + files legs of bench.py
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_cli_gpu.py tests/test_raster_gpu.py -x -q 2>&1 | tail -3 | tee gpurun_out/r06/files_tests.log
for th in 16 32 64; do
  python - $th <<'PY' 2> gpurun_out/r06/cli_$1.err | tee -a gpurun_out/r06/files_sweep.log
import sys, time, os, shutil, contextlib, io
sys.path.insert(0, os.getcwd())
import generate_vessel_graph
th = sys.argv[1]
for rep in range(2):
    shutil.rmtree("/dev/shm/octa_cli_out", ignore_errors=True)
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        generate_vessel_graph.main(["--config_file", "docker/vessel_graph_gen_docker_config.yml", "--num_samples", "4096", "--labels", "--seed", "7000000", "--threads", th, "--output.directory", "/dev/shm/octa_cli_out"])
    dt = time.time() - t
    print(f"threads {th} rep {rep}: {len(os.listdir('/dev/shm/octa_cli_out'))} dirs, {4096 / dt:.1f} triples/s", flush=True)
shutil.rmtree("/dev/shm/octa_cli_out", ignore_errors=True)
PY
done
timeout 900 python bench.py --no-train --no-pmc --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r06/bench_files.json 2> gpurun_out/r06/bench_files.err
python -c "
import json; d=json.load(open('gpurun_out/r06/bench_files.json')); print('headline', d['value'], 'kernel', d['roofline']['avg_launch_ms'], 'files', d['files']['value'], d['files']['write_seconds'], 'cli', d['files']['cli_pipelined']['value'], d['slot_cycle'])" | tee -a gpurun_out/r06/files_sweep.log
