import sys, time, os, shutil, contextlib, io
sys.path.insert(0, os.getcwd())
import generate_vessel_graph
n = 8192
best = 0
for rep in range(3):
    shutil.rmtree("/dev/shm/octa_cli_out", ignore_errors=True)
    t = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        generate_vessel_graph.main(["--config_file", "docker/vessel_graph_gen_docker_config.yml", "--num_samples", str(n), "--labels", "--seed", "7000000", "--output.directory", "/dev/shm/octa_cli_out"])
    dt = time.time() - t
    print(f"rep {rep}: {len(os.listdir('/dev/shm/octa_cli_out'))} dirs, {n / dt:.1f} triples/s", flush=True)
shutil.rmtree("/dev/shm/octa_cli_out", ignore_errors=True)
