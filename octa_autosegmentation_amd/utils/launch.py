"""One process per GPU, started by the program itself: `bench.py --gpus N` (and any other entry point that takes a device count) run as
plain `python` re-executes itself under `torch.distributed.run --standalone --nproc-per-node N` on 127.0.0.1, one rank per GPU over RCCL.
The reference fans its generator out over a process pool from inside the CLI the same way (generate_vessel_graph.py:112-129: the user never
starts the workers). Under torchrun (WORLD_SIZE set by the launcher) nothing is re-executed."""
import os
import subprocess
import sys


class LaunchError(RuntimeError):
    pass


def visible_gpus():
    """GPUs this process can open. Counted in a child process when torch has not touched the device yet: the parent of N ranks must not
    hold a context on GPU 0."""
    out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count() if torch.cuda.is_available() else 0)"],
                         capture_output=True, text=True)
    try:
        return int(out.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        raise LaunchError(f"could not count the visible GPUs: {out.stderr.strip()[-400:]}")


def needs_self_launch(n_gpus, environ=None):
    """True when `--gpus N` (N > 1) was given to a process no launcher started."""
    environ = os.environ if environ is None else environ
    return int(n_gpus) > 1 and "WORLD_SIZE" not in environ


def check_world(n_gpus, environ=None):
    """Under a launcher the world size IS the number of ranks; `--gpus` has to say the same (the driver passes both)."""
    environ = os.environ if environ is None else environ
    world = int(environ.get("WORLD_SIZE", "1"))
    if int(n_gpus) != world and not (int(n_gpus) == 1 and "WORLD_SIZE" not in environ):
        raise LaunchError(f"--gpus {n_gpus} but the launcher started WORLD_SIZE={world} ranks: start it as `python bench.py --gpus N` "
                          f"(it launches its own ranks) or under `torch.distributed.run --nproc-per-node N` with the same N")
    return world


def self_launch(script, argv, n_gpus, need_devices=True, n_visible=None, timeout=None):
    """Re-execute `script argv` as n_gpus ranks of one node and return the exit code. Fails BEFORE starting anything when fewer than
    n_gpus devices are visible (need_devices=False: CPU self-tests over gloo). stdout / stderr of the ranks are this process's own, so the
    ONE JSON line rank 0 prints is this command's output."""
    n_gpus = int(n_gpus)
    if need_devices:
        have = visible_gpus() if n_visible is None else int(n_visible)
        if have < n_gpus:
            raise LaunchError(f"--gpus {n_gpus}: only {have} GPU(s) visible to this process (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES "
                              f"= {os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('ROCR_VISIBLE_DEVICES', 'unset'))}); nothing was started")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL's intra-node transport on these hosts
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_gpus)))
    env["OCTA_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n_gpus}", script] + list(argv)
    return subprocess.run(cmd, env=env, timeout=timeout).returncode
