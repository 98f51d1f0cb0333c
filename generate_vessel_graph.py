"""Drop-in for the reference's generate_vessel_graph.py on MI355X.

Same flags (`--config_file`, `--num_samples`, `--debug`, `--threads`, dotted `--A.B.c value` overrides) and the
same outputs per sample under `<output.directory>/<YYYYmmdd_HHMMSS>_<uuid4>/`: `config.yml`, `<name>.csv`
(`node1,node2,radius`), `art_ven_img_gray.png` (+ `art_ven_img_gray.npy` for save_3D_volumes: npy), written
from GPU results: all samples are simulated in lock-step batches by the HIP simulator and rasterised by the HIP
rasteriser. Additive flags: `--seed S` (sample k uses random.seed(S+k); np.random.seed(S+k); the reference never
seeds), `--batch B` (samples per GPU batch, default 512), `--device N` / `--devices 0-7` (one group of `--inflight` generator threads per
listed GPU, all taking launches from one queue; the reference fans samples out over every worker of the machine,
generate_vessel_graph.py:112-129), `--labels` (also write `<name>_label.png`, the 1216x1216 binarised label
visualize_vessel_graphs.py --binarize would render from the CSV: complete triples in one pass). Under `torch.distributed.run` (one
process per GPU) every rank takes every WORLD_SIZE-th launch on GPU LOCAL_RANK. Sample k is seeded by `--seed` + k wherever it runs.
CSV text and PNG files are formatted / encoded natively by a pool of host threads (`--threads`) while the GPU simulates
the next batch.
"""
import argparse
import os
import random
import sys
import time
import warnings
from datetime import datetime
from uuid import uuid4

import numpy as np
import yaml


def main(argv=None):
    parser = argparse.ArgumentParser(description='')
    parser.add_argument('--config_file', type=str, required=True)
    parser.add_argument('--num_samples', type=int, default=1)
    parser.add_argument('--debug', action="store_true")
    parser.add_argument('--threads', type=int, default=-1, help="host threads that format and write the files (samples run on the GPU); default 16: more "
                        "writers slow the simulator's host-side LAPACK service down (measured: 4 -> 350, 8 -> 501, 12 -> 512, 16 -> 561, 32 -> 435 triples/s)")
    parser.add_argument('--labels', action="store_true", help="also write <name>_label.png (1216x1216, binarised)")
    parser.add_argument('--seed', type=int, default=None)
    parser.add_argument('--batch', type=int, default=512, help="samples per launch of the simulator (512 = every workgroup slot of an MI355X)")
    parser.add_argument('--inflight', type=int, default=2, help="generator threads (own simulator state, rasteriser scratch and HIP stream each): one batch is on the GPU "
                        "while the other threads copy theirs out and hand them to the file writers")
    parser.add_argument('--device', type=int, default=0)
    parser.add_argument('--devices', type=str, default=None, help="GPUs of this process, e.g. 0-7 / 0,2,5 / all: one generator group per listed GPU (overrides --device)")
    args, unknown = parser.parse_known_args(argv)
    if args.debug:
        warnings.filterwarnings('error')
    assert os.path.isfile(args.config_file), f"Error: Your provided config path {args.config_file} does not exist!"
    with open(os.path.abspath(args.config_file), "r") as f:
        config = yaml.safe_load(f)

    from octa_autosegmentation_amd import graph_io, pipeline
    from octa_autosegmentation_amd.utils.config_overrides import apply_cli_overrides_from_unknown_args
    from octa_autosegmentation_amd.vessel_graph_generation import tree2img
    apply_cli_overrides_from_unknown_args(config, unknown)
    out_cfg = config['output']
    assert out_cfg.get('save_3D_volumes') in [None, 'npy', 'nifti'], \
        f"Your provided option {out_cfg.get('save_3D_volumes')} for 'save_3D_volumes' does not exist. Choose one of 'null', 'npy' or 'nifti'."

    import torch
    from octa_autosegmentation_amd.output_files import SampleFileWriter
    from octa_autosegmentation_amd.utils import sharding
    # placement: ranks of a torchrun job take every WORLD_SIZE-th launch on their own GPU; inside a process one generator group per device
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and args.devices is None and "LOCAL_RANK" in os.environ:
        devices = [int(os.environ["LOCAL_RANK"])]
    else:
        devices = sharding.parse_devices(args.devices, torch.cuda.device_count()) or [args.device]
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world))) if world > 1 else 1
    budget = sharding.apply_host_budget(local_world=max(local_world, len(set(devices))), generator_threads=args.inflight * len(devices),
                                        set_affinity=world > 1)
    torch.cuda.set_device(devices[0])
    if args.seed is None and world > 1:
        raise SystemExit("generate_vessel_graph.py under torch.distributed.run needs --seed (every rank must derive the same seeds)")
    seed0 = args.seed if args.seed is not None else random.SystemRandom().randrange(0, 2 ** 31 - args.num_samples - 1)
    writer = SampleFileWriter(args.threads if args.threads > 0 else budget["writers"])
    # Batches are independent: `--inflight` generator threads (own simulator state, rasteriser scratch and HIP stream each) keep
    # that many launches of the persistent kernel on the GPU while this thread hands finished batches to the file writers -- the
    # reference's process pool over samples (generate_vessel_graph.py:112-129) with the roles of CPU and GPU exchanged.
    import queue
    import threading
    sys.setswitchinterval(0.0005)      # generator threads, the submitting thread and the writers' glue share the GIL: hand it over in 0.5 ms slices, not 5 ms ones
    plan = sharding.plan_batches(args.num_samples, args.batch, rank, world)
    n_mine = sum(n for _, n in plan)
    n_fly = max(1, min(args.inflight, len(plan)))
    todo = queue.Queue()
    for item in plan:
        todo.put(item)
    finished = queue.Queue(maxsize=max(2, len(devices)))                     # bounds the host memory held by batches waiting for the writers
    # pinned staging sets (output_files.HostStaging): one per batch between its copy-out and the end of its file writes -- a batch being
    # written, the ones waiting in `finished`, one per generator thread being filled
    from octa_autosegmentation_amd.output_files import HostStaging
    staging = queue.Queue()
    for _ in range(args.inflight * len(devices) + max(2, len(devices)) + 1):
        staging.put(HostStaging())
    failure = []
    stop = threading.Event()
    timing = os.environ.get("OCTA_CLI_TIMING", "0") == "1"
    # At most one GPU's worth of samples (512 workgroup slots on an MI355X: two samples per CU; OCTA_GPU_SLOTS) is simulated and rasterised at a time; batches beyond that
    # wait here while their threads' finished batches are copied out and handed to the writers. Two reasons (6144 samples in batches
    # of 512, MI355X): persistent kernels of several launches resident at once double the working set that already misses L2, and the
    # rasteriser's launch sequence reads sizes back, i.e. waits behind another thread's persistent kernel for as long as that runs.
    class _SlotGate:
        def __init__(self, capacity):
            self.capacity, self.used, self.cv = capacity, 0, threading.Condition()

        def acquire(self, n):
            with self.cv:
                while self.used and self.used + n > self.capacity:
                    self.cv.wait()
                self.used += n

        def release(self, n):
            with self.cv:
                self.used -= n
                self.cv.notify_all()

    gates = {d: _SlotGate(int(os.environ.get("OCTA_GPU_SLOTS", "512"))) for d in set(devices)}      # one gate per GPU

    # Launches that fill the GPU (>= the slot count) are ordered like bench.py's (round 6): ONE persistent kernel at a time through a
    # pipeline.SimGate per device, a launch's rasterisation on the GPU together with the NEXT launch's kernel, their order kept on the device
    # (csrc/order.hip). Smaller batches keep the slot gate: their whole generate() call holds its share of the GPU.
    sim_gates = {d: pipeline.SimGate() for d in set(devices)}
    use_sim_gate = True

    def generate_batches(dev):
        gens = {}
        gpu_gate = gates[dev]
        try:
            torch.cuda.set_device(dev)
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                while not stop.is_set():
                    try:
                        start, B = todo.get_nowait()
                    except queue.Empty:
                        break
                    if B not in gens:
                        gens[B] = pipeline.TripleGenerator(config, B)
                        if use_sim_gate and B >= gpu_gate.capacity and n_fly > 1:
                            gens[B].sim_gate = sim_gates[dev]
                    seeds = np.arange(seed0 + start, seed0 + start + B, dtype=np.int64).astype(np.uint32)
                    t_a = time.time()
                    if gens[B].sim_gate is not None:
                        out = gens[B].generate(seeds, want_label=args.labels)
                        stream.synchronize()
                    else:
                        gpu_gate.acquire(B)
                        try:
                            out = gens[B].generate(seeds, want_label=args.labels)
                            stream.synchronize()
                        finally:
                            gpu_gate.release(B)
                    res = out["result"]
                    t_b = time.time()
                    # copy-out into a pinned staging set that is this batch's until its files are written (round 6: pageable .cpu() copies of
                    # ~0.5 GB per batch; the label leaves the GPU as mode "1" rows, an eighth of its bytes)
                    stage = staging.get()
                    want_3d = bool(out_cfg.get("save_3D_volumes"))
                    host = stage.fetch(images=out["image"], labels=tree2img.pack_label_bits_device(out["label"]) if args.labels else None,
                                       edges=res.d_edges if (res.d_edges is not None and out_cfg.get('save_trees', True) and not want_3d) else None)
                    images, labels = host["images"], host["labels"]
                    if host["edges"] is not None:
                        res._edges = host["edges"]
                    t_c = time.time()
                    vols = None
                    if out_cfg.get("save_3D_volumes"):
                        shape = np.array([config['Greenhouse']['SimulationSpace'][a] for a in ("no_voxel_x", "no_voxel_y", "no_voxel_z")])
                        vol_dim = [int(d) for d in shape * out_cfg['image_scale_factor']]
                        vols = []
                        for k in range(B):
                            d_edges = torch.from_numpy(np.ascontiguousarray(res.sample_edges(k))).to(torch.device("cuda", dev))
                            na = int(res.n_art[k])
                            v = tree2img.voxelize_edges_device(d_edges, np.array([0, na, len(d_edges)]), vol_dim)
                            vols.append(torch.maximum(v[0], v[1]).cpu().numpy().astype(np.uint8))
                    finished.put((B, res, images, labels, vols, stage, int(out["label"].shape[2]) if args.labels else 0))
                    if timing:
                        print(f"[cli timing] generator: generate {t_b - t_a:.3f} s (simulator {out['wall']['sim_run_s']:.3f}), wait + copy out {t_c - t_b:.3f}, "
                              f"hand over {time.time() - t_c:.3f}", file=sys.stderr, flush=True)
        except BaseException as e:                            # noqa: BLE001 -- re-raised by the main thread
            failure.append(e)
            stop.set()
        finally:
            for g in gens.values():
                g.close()
            finished.put(None)

    threads = [threading.Thread(target=generate_batches, args=(d,), name=f"octa-generator-{gi}-{i}") for gi, d in enumerate(devices) for i in range(n_fly)]
    n_fly = len(threads)
    for t in threads:
        t.start()
    done = 0
    alive = n_fly
    try:
        while alive:
            item = finished.get()
            if item is None:
                alive -= 1
                continue
            B, res, images, labels, vols, stage, gen_label_width = item
            t_a = time.time()
            writer.wait()                                  # the previous batch's files (written while the next ones were simulated)
            t_b = time.time()
            stamp = datetime.now().strftime('%Y%m%d_%H%M%S')
            root_dir = os.path.abspath(out_cfg['directory'])
            out_dirs = [os.path.join(root_dir, stamp + "_" + str(uuid4())) for _ in range(B)]
            names = [os.path.basename(d) for d in out_dirs]
            if vols is None:
                # one native call per batch (csrc/fileio.cpp octa_write_sample_files): its threads take the samples from a counter
                writer.submit_batch(out_dirs, names, edges=res.edges if out_cfg.get('save_trees', True) else None, edge_off=res.edge_off,
                                    images=images if out_cfg.get("save_2D_image", True) else None, label_bits=labels,
                                    label_width=gen_label_width if labels is not None else None, config=config, on_done=lambda st=stage: staging.put(st))
            else:
                if labels is not None:                         # the per-sample writers take one byte per pixel
                    labels = np.unpackbits(labels, axis=2)[:, :, :gen_label_width] * np.uint8(255)
                images = np.array(images)
                staging.put(stage)                             # everything this path hands on is a copy
                for k in range(B):
                    writer.submit(out_dirs[k], names[k], edges=res.sample_edges(k) if out_cfg.get('save_trees', True) else None,
                                  image=images[k] if out_cfg.get("save_2D_image", True) else None,
                                  label_bits=labels[k] if labels is not None else None, config=config, volume=vols[k],
                                  volume_format=out_cfg.get("save_3D_volumes") or "npy")
            done += B
            if timing:
                print(f"[cli timing] main: waited {t_b - t_a:.3f} s for the previous batch's files, submitted {B} samples in {time.time() - t_b:.3f}", file=sys.stderr, flush=True)
            print(f"generated {done}/{n_mine} vessel graphs" + (f" (rank {rank} of {world})" if world > 1 else ""))
    finally:
        stop.set()                                         # after a failure here or in a generator: let the other threads run out
        while alive:
            if finished.get() is None:
                alive -= 1
        for t in threads:
            t.join()
        writer.close()                                     # a failed write raises here (the reference drops worker exceptions)
    if failure:
        raise RuntimeError("a generator thread failed") from failure[0]


if __name__ == '__main__':
    main()
