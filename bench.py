"""bench.py -- headline benchmark: synthetic OCTA triples per second on MI355X.

A step = one pass of the hot path over one batch of seeded samples with the reference's own generator config
(docker/vessel_graph_gen_docker_config.yml): space-colonisation simulation of B vessel graphs (HIP), 304x304
arterial/venous rasterisation and max-combine, 1216x1216 label rasterisation + Floyd-Steinberg binarisation.
Workload = BASELINE.json configs[1] (128-sample batch, rasterise to 1216x1216). `value` counts triples complete in HBM
(edge list, image and label are device tensors: the edge list is exported by a kernel and read by the rasteriser where it is);
`value_with_csv` is the rate of the drop-in CLI writing the reference's per-sample FILES (graph CSV + image PNG + label PNG: the
triple as SURVEY.md 8d defines it), measured by the `files` leg of the same run.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
Multi-GPU (driver-launched with torch.distributed.run): samples are independent, every rank generates its own batch with
its own seeds; no data-path collective (weak scaling).

Every number quoted in DESIGN.md section 5 is a field of the JSON line this prints (or of profiles/).
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

# one HIP hardware queue per step in flight (the runtime's default of 4 would make streams share queues)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("OCTA_STRICT", "1")      # a measured pass that leaves the hand-written kernels for the vendor libraries is an error (models/networks.py)

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_SAMPLE = 84e6       # point traffic of one full-length sample (all iterations)
RASTER_BYTES_1216 = 2.2e6          # edges * 56 B + 1216 * 1216 B per label image
RASTER_BYTES_304 = 0.82e6
UNET_TFLOP_PER_IMAGE = 2.0         # forward 0.666 TFLOP x 3
UNET_MIN_HBM_GB_PER_IMAGE = 6.0    # activations written once / read once, norm + activation fused
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8.0 TB/s spec
MFMA_BF16_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16
VOXEL_BYTES_CLI = 1216 * 1216 * 16 * 2      # the volume the CLI shape names, uint16 (47 MB)
G_TFLOP_FWD_304 = 0.178            # resnetGenerator9 forward at 1x304x304
D_TFLOP_FWD_304 = 0.018            # patchGAN70x70 forward at 1x304x304
N_CUS = 256          # MI355X; main() replaces it by the device's own count
PMC_SUMMARY = os.path.join("profiles", "r05_bench_pmc_summary.csv")     # this round's counter passes (tools/profile_bench.sh); used only when the live passes fail


def pmc_traffic_per_launch(kernel_name, grid_threads=None):
    """HBM-side bytes per launch of `kernel_name` from the committed rocprofv3 counter passes (PMC_SUMMARY: FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc passes of this bench by tools/profile_bench.sh, values in KB; one row per kernel and launch
    size, `[grid <threads>]`). Correction per MI355X_MICROARCH.md (HBM): FETCH_SIZE counts 128-B requests as 64 B on gfx950, so it is
    doubled; WRITE_SIZE is taken as reported. Returns (bytes, file, exact): exact = the file holds a row for launches of
    `grid_threads` threads; otherwise the rows of another launch size are returned for the caller to scale. None if missing."""
    for cand in (PMC_SUMMARY,):
        path = os.path.join(ROOT, cand)
        if not os.path.exists(path):
            continue
        import csv
        kb, kb_exact = {}, {}
        with open(path) as f:
            for r in csv.DictReader(f):
                if kernel_name in r["kernel"]:
                    if grid_threads is not None and f"[grid {grid_threads}]" in r["kernel"]:
                        kb_exact[r["counter"]] = float(r["avg_KB_per_launch"])
                    elif "[grid" not in r["kernel"] or "[grid 32768]" in r["kernel"]:
                        kb[r["counter"]] = float(r["avg_KB_per_launch"])       # 128-sample launches (all the older files hold)
        if "FETCH_SIZE" in kb_exact and "WRITE_SIZE" in kb_exact:
            return (2.0 * kb_exact["FETCH_SIZE"] + kb_exact["WRITE_SIZE"]) * 1024.0, cand, True
        if "FETCH_SIZE" in kb and "WRITE_SIZE" in kb:
            return (2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0, cand, False
    return None, None, False


def pmc_child(batch):
    """Hidden mode (`bench.py --pmc-child`): one warm-up and one measured 128-sample launch of the simulator with the GPU to itself;
    run under `rocprofv3 --pmc <counter>` by pmc_traffic_live()."""
    import torch
    from octa_autosegmentation_amd import pipeline
    from octa_autosegmentation_amd.utils import sharding
    torch.cuda.set_device(0)
    gen = pipeline.TripleGenerator(load_config(), batch)       # simulator + both rasterisations + dither: the rasteriser's kernels are counted too
    for i in range(2):
        out = gen.generate(sharding.rank_seeds(0, 800 + i, batch))
        torch.cuda.synchronize()
        assert int(out["result"].stats[:, 0].max()) == 0
    gen.close()


RASTER_KERNELS = ("raster_meta_kernel", "raster_scan", "raster_tess_kernel", "raster_render_kernel", "fs_dither", "read_back_kernel", "max_u8")


def pmc_traffic_live(kernel_name, batch, extra=None):
    """HBM-side bytes per launch of `kernel_name`, measured IN THIS RUN when rocprofv3 is on PATH: one separate counter pass per
    counter (FETCH_SIZE, WRITE_SIZE; --pmc alone, no tracing -- MI355X_MICROARCH.md's recipe) over `bench.py --pmc-child` (two
    generate() calls of one `batch`-sample batch: simulator launch + rasteriser launch sequences + dither). Same correction as the
    file-based figure: 2 x FETCH_SIZE + WRITE_SIZE, counters in KB. Returns (bytes per launch, description) or (None, reason).
    `extra` (a dict) receives, per rasteriser kernel, its launches and the SUM of its traffic over one generate() call."""
    import csv
    import glob
    import subprocess
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not on PATH"
    kb = {}
    tmp = tempfile.mkdtemp(prefix="octa_pmc_", dir="/tmp")
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, c)
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run([exe, "--pmc", c, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--pmc-child",
                                "--batch", str(batch)], cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=420)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {c} failed (rc {r.returncode}): {r.stdout[-300:]}"
            tot, n = 0.0, 0
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") != c:
                        continue
                    kn = row.get("Kernel_Name", "")
                    if kernel_name in kn:
                        tot += float(row["Counter_Value"]); n += 1
                    elif extra is not None:
                        for rk in RASTER_KERNELS:
                            if rk in kn:
                                e = extra.setdefault(f"{rk} [grid {row.get('Grid_Size', '?')}]", {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "launches": 0})
                                e[c] += float(row["Counter_Value"]) / 2.0          # the child runs generate() twice
                                if c == "FETCH_SIZE":
                                    e["launches"] += 0.5
                                break
            if n == 0:
                return None, f"no {c} rows for {kernel_name}"
            kb[c] = tot / n
        return (2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, "
                                                                      f"2 launches of {batch} samples each, GPU otherwise idle)")
    except Exception as e:  # noqa: BLE001 -- the counters are a report, never a reason to lose the bench line
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def load_config():
    from octa_autosegmentation_amd.utils import configs
    return configs.load_generator_config()


def _oracle_sample(args):
    cfg, seed = args
    from oracle import octa_oracle, sim_oracle
    from octa_autosegmentation_amd import graph_io
    e, info = sim_oracle.simulate(cfg, seed)
    na = info["n_art_edges"]
    np.maximum(octa_oracle.rasterize(e[:na], [304, 304]), octa_oracle.rasterize(e[na:], [304, 304]))
    octa_oracle.fs_dither(octa_oracle.rasterize(graph_io.edges_as_read_back(e), [1216, 1216]))
    return len(e)


def cpu_baseline(cfg, all_cores=False):
    """The oracle (CPU restatement of the reference, oracle/) on the GPU box's host cores: ONE full-length sample per core on
    all cores at once (the reference's own parallelism: one sample per pool worker, generate_vessel_graph.py:112-129), and the
    single-core figure. Bounded: one wave of samples (about 5-10 s) + one more sample."""
    from multiprocessing import get_context
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # 64 workers, not one per visible core: measured on the 256-thread host of the GPU box (round 3), 256 workers give 1.1 samples/s
    # against 2.2 with 64 -- the restatement's brute-force neighbour queries are memory-bound and the wave of 256 takes 230 s, far
    # beyond the bounded 10-30 s sample this leg is allowed; `host_cores_visible` reports what the box has
    workers = min(cores, 64)
    t0 = time.time()
    _oracle_sample((cfg, 900))
    single = 1.0 / (time.time() - t0)
    t0 = time.time()
    with get_context("spawn").Pool(workers) as pool:
        pool.map(_oracle_sample, [(cfg, 901 + k) for k in range(workers)])
    dt = time.time() - t0
    all_info = {"value": None, "cores": cores, "note": "not run by default (one worker per visible core takes ~230 s on the 256-thread host); measured with --cpu-all-cores: 1.13 samples/s "
                                                       "with 256 workers against 2.40 with 64 on the same host (round 6; round 3: 1.1 against 2.2)"}
    if all_cores and cores > workers:
        t1 = time.time()
        with get_context("spawn").Pool(cores) as pool:
            pool.map(_oracle_sample, [(cfg, 2000 + k) for k in range(cores)])
        all_info = {"value": cores / (time.time() - t1), "cores": cores, "note": "one full-length sample per visible core, all at once"}
    return {"value": workers / dt, "unit": "samples/s", "cores": workers, "kind": "port", "host_cores_visible": cores,
            "single_core_value": single, "all_cores": all_info,
            "sample": f"{workers} full-length samples (I=100+150, N=2000) incl. 304x304 image + 1216x1216 label, one per core on {workers} cores "
                      f"at once (pool start-up included), oracle/ C++; single core: one sample"}


def cpu_unet_step():
    """plain torch DynUNet-S fp32 training step on the host CPU, B = 1 at 1x1216x1216."""
    import torch
    from octa_autosegmentation_amd.models import networks
    torch.manual_seed(0)
    net = networks.DynUNet()
    networks.init_weights(net, "kaiming")
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, betas=(0.5, 0.999))
    x, y = torch.rand(1, 1, 1216, 1216), (torch.rand(1, 1, 1216, 1216) > 0.8).float()
    from octa_autosegmentation_amd.models.losses import DiceBCELoss
    loss_f = DiceBCELoss(True)
    def step():
        opt.zero_grad()
        loss_f(net(x), y).backward()
        opt.step()

    t0 = time.time()
    step()                                   # warm-up: first-call overheads (thread pools, oneDNN primitive creation)
    cold = time.time() - t0
    times = []
    for _ in range(1):                        # one timed step (13 s on the 128-core host): the leg stays within its 30 s
        t0 = time.time()
        step()
        times.append(time.time() - t0)
    dt = sum(times) / len(times)
    return {"value": 1.0 / dt, "unit": "imgs/s", "cores": torch.get_num_threads(), "kind": "plain torch fp32 on the CPU device",
            "cold_first_step_s": cold, "step_s": times,
            "sample": "DynUNet-S training steps, B=1, 1x1216x1216: one warm-up step, then one timed step"}


def voxel_leg(cfg, dev, n_vol=8):
    """the 3-D voxeliser at the CLI's shape [1216, 1216, 16] (padded to z = 53 as the reference pads,
    tree2img.py:206-211) on the edge lists of freshly simulated full-length samples; HIP events on the launch stream."""
    import torch
    from octa_autosegmentation_amd.utils import sharding
    from octa_autosegmentation_amd.vessel_graph_generation import greenhouse, tree2img
    sim = greenhouse.BatchSimulator(cfg, n_vol)
    res = sim.run(sharding.rank_seeds(0, 970, n_vol))
    d_edges = res.d_edges if res.d_edges is not None else torch.from_numpy(res.edges).to(dev)
    off = np.asarray(res.edge_off, dtype=np.int64)
    dims = [1216, 1216, 16]
    vol = tree2img.voxelize_edges_device(d_edges, off, dims)            # warm-up (scratch allocation)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    e0.record()
    for _ in range(reps):
        vol = tree2img.voxelize_edges_device(d_edges, off, dims)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / (reps * n_vol)
    padded = int(vol.shape[1]) * int(vol.shape[2]) * int(vol.shape[3]) * 2
    sim.close()
    gbs = VOXEL_BYTES_CLI / (ms * 1e-3) / 1e9
    return {"kernel": "octa_voxelize_3d (voxel_edges_kernel + voxel_widen_kernel)", "bound": "hbm", "ms_per_volume": ms, "volumes": n_vol,
            "edges_per_volume": float(off[-1]) / n_vol, "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "algorithmic_bytes_per_volume": VOXEL_BYTES_CLI, "padded_volume_bytes": padded,
            "achieved_padded": padded / (ms * 1e-3) / 1e9,
            "note": "1216 x 1216 x 16 (padded to z = 53: 157 MB written per volume); the per-edge box walk in double arithmetic, not the streaming part, "
                    "dominates (DESIGN.md 4.2b); the reference takes 58.7 s per volume"}


def gan_networks_leg(dev, batch=4, steps=20, warmup=5):
    """forward passes of the GAN's generator (resnetGenerator9, 0.178 TFLOP at 304^2) and discriminator (patchGAN70x70,
    0.018 TFLOP) under bf16 autocast on the MFMA / streaming kernels: TFLOP/s against the dense bf16 peak."""
    import torch
    from octa_autosegmentation_amd.models import networks
    out = {}
    x = torch.rand(batch, 1, 304, 304, device=dev)
    for name, make, tf in (("resnetGenerator9", networks.resnetGenerator9, G_TFLOP_FWD_304), ("patchGAN70x70", networks.patchGAN70x70, D_TFLOP_FWD_304)):
        torch.manual_seed(0)
        net = make().to(dev).eval()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
            for _ in range(warmup):
                net(x)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(steps):
                net(x)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / steps
        tfs = tf * batch / (ms * 1e-3)
        out[name] = {"ms_per_forward": ms, "batch": batch, "imgs_per_s": batch / (ms * 1e-3), "bound": "mfma", "achieved": tfs, "peak": MFMA_BF16_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": tfs / MFMA_BF16_PEAK_TFLOPS}
    out["note"] = ("forward only, B = 4 at 1 x 304 x 304, bf16 autocast; the 76 x 76 residual stages of the generator are launch- / latency-bound at this "
                   "batch (DESIGN.md 4.2f), the discriminator is dominated by its one-channel stem / head and the blur kernels")
    return out


def allreduce_leg(tr, dev, dist, world, reps=20):
    """the ONE exchange of the data-parallel step -- an RCCL all-reduce over the flat fp32 gradient buffer of DynUNet-S
    (7.37 M parameters, 29.5 MB) -- timed on its own with HIP events on the current stream, `reps` times back to back. With several ranks
    it runs on the job's process group (barrier first, MAX over ranks); with one rank a one-rank RCCL group is created for the measurement
    (communicator set-up, the collective's launch and its device-side copy are real; there is no wire)."""
    import torch
    import torch.distributed as tdist
    n = sum(p.numel() for p in tr.model.parameters() if p.requires_grad)
    flat = torch.zeros(n, dtype=torch.float32, device=dev)
    own_group = False
    try:
        if dist is None:
            if not tdist.is_available():
                return {"value": None, "note": "torch.distributed unavailable"}
            if not tdist.is_initialized():
                import socket
                with socket.socket() as sk:
                    sk.bind(("127.0.0.1", 0))
                    port = sk.getsockname()[1]
                tdist.init_process_group(backend="nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
                own_group = True
        for _ in range(3):
            tdist.all_reduce(flat)
        torch.cuda.synchronize()
        if world > 1:
            tdist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            tdist.all_reduce(flat)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        if world > 1:
            t = torch.tensor([ms], dtype=torch.float64, device=dev)
            tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
            ms = float(t.item())
        bus = 2.0 * (world - 1) / max(world, 1) * n * 4 / (ms * 1e-3) / 1e9 if world > 1 else None
        return {"value": ms, "unit": "ms", "ranks": world, "bytes": n * 4, "bus_GBps": bus,
                "note": ("RCCL all-reduce(sum) of the flat fp32 gradient arena, once per optimiser step (models/base_model_abc.py: GradArena)"
                         + ("" if world > 1 else "; ONE rank: a one-rank RCCL group made for this measurement -- no wire, the figure is the collective's launch + device-side "
                                                 "cost and becomes the real exchange when the driver runs --gpus N"))}
    except Exception as e:  # noqa: BLE001 -- a report, never a reason to lose the bench line
        return {"value": None, "note": f"{type(e).__name__}: {e}"}
    finally:
        if own_group:
            tdist.destroy_process_group()


def unet_train_bench(dev, batch, dist, world, steps=20, warmup=4, with_allreduce=False):
    import torch
    from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
    cfg = {"General": {"amp": True, "model": {"name": "DynUNet", "spatial_dims": 2, "in_channels": 1, "out_channels": 1,
                                               "kernel_size": [3, 3, 3, 3, 3], "strides": [1, 2, 2, 2, 1],
                                               "upsample_kernel_size": [1, 2, 2, 2, 1]}},
           "Train": {"lr": 1e-4, "loss": "DiceBCELoss", "epochs": 30, "epochs_decay": 10}}
    torch.manual_seed(0)
    tr = SegmentationTrainer(cfg, dev)
    x = torch.rand(batch, 1, 1216, 1216, device=dev)
    y = (torch.rand(batch, 1, 1216, 1216, device=dev) > 0.8).float()
    for _ in range(warmup):
        tr.perform_training_step({"image": x, "label": y})
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.time()
    for _ in range(steps):
        tr.perform_training_step({"image": x, "label": y})
    torch.cuda.synchronize()
    dt = time.time() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ips = world * batch * steps / dt
    per_gpu = ips / world
    allreduce = allreduce_leg(tr, dev, dist, world) if with_allreduce else None
    return {"metric": "DynUNet-S training imgs/s @1x1216x1216", "value": ips, "unit": "imgs/s", "dtype": "bf16",
            "batch_per_gpu": batch, "ms_per_step": dt / steps * 1e3, "tflops": UNET_TFLOP_PER_IMAGE * ips,
            "allreduce_ms_per_step": allreduce,
            "roofline": {"mfma": {"achieved": UNET_TFLOP_PER_IMAGE * per_gpu, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                  "frac": UNET_TFLOP_PER_IMAGE * per_gpu / MFMA_BF16_PEAK_TFLOPS},
                         "hbm": {"achieved": UNET_MIN_HBM_GB_PER_IMAGE * per_gpu, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": UNET_MIN_HBM_GB_PER_IMAGE * per_gpu / HBM_PEAK_GBS,
                                 "note": "algorithmic minimum traffic (6 GB per training image with norm + activation fused), not measured traffic"},
                         "binding": "mfma"},
            "parity": "bf16 activations: logits / gradients within bf16's error budget of an fp32 run on the same bf16-rounded weights "
                      "(tests/test_fullsize_gpu.py); north_star's 1e-4 holds for the fp32 path (tests/test_models_gpu.py); DynUNet / DiceLoss are "
                      "restated from MONAI's documentation (MONAI absent: parity with MONAI itself unpinned)",
            "implementation": "channels-last bf16 on hand-written HIP kernels: MFMA 3x3 conv forward / data gradient / weight "
                              "gradient (csrc/conv.hip; skip-connection gradients in the data-gradient epilogue), NHWC InstanceNorm+LeakyReLU "
                              "(csrc/norm.hip; the last one fused with the 1x1 output convolution), one-launch weight packing; the 1x1 transposed "
                              "convolution at the bottleneck on the same MFMA kernels with a one-tap mask (round 4: no vendor-library kernel in the step); "
                              "flat RCCL gradient all-reduce"}


def files_leg(gen, stream, seeds, threads=None):
    """One batch through generate -> native CSV / PNG writers: the reference's per-sample files (generate_vessel_graph.py:43-86 +
    visualize_vessel_graphs.py:95-101) on disk. Returns on-disk triples per second, generation included."""
    import torch
    from octa_autosegmentation_amd.output_files import SampleFileWriter, default_threads
    out_root = tempfile.mkdtemp(prefix="octa_bench_files_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    writer = SampleFileWriter(threads)
    try:
        torch.cuda.synchronize()
        t0 = time.time()
        with torch.cuda.stream(stream):
            out = gen.generate(seeds)
            images = out["image"].cpu().numpy()
            labels = out["label"].cpu().numpy()
        t_gen = time.time() - t0
        res = out["result"]
        names = [f"sample_{int(s_)}" for s_ in seeds]
        writer.submit_batch([os.path.join(out_root, n_) for n_ in names], names, edges=res.edges, edge_off=res.edge_off, images=images, label_bits=labels)
        writer.wait()
        dt = time.time() - t0
        nbytes = sum(os.path.getsize(os.path.join(d, f)) for d, _, fs in os.walk(out_root) for f in fs)
        # the reference's entry point itself, pipelined: generate_vessel_graph.py --labels simulates and rasterises one 512-sample batch
        # while the previous one is copied out and written
        shutil.rmtree(out_root, ignore_errors=True)
        os.makedirs(out_root, exist_ok=True)
        import contextlib
        import io
        import generate_vessel_graph
        from octa_autosegmentation_amd.utils import configs
        n_cli = 8192                 # round 6: 16 launches of 512 (4096 until round 5: the set-up of the generator threads was a fifth of the leg)
        t1 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            generate_vessel_graph.main(["--config_file", configs.GENERATOR_CONFIG, "--num_samples", str(n_cli), "--labels", "--seed", "7000000", "--output.directory", out_root])
        dt_cli = time.time() - t1
        n_dirs = len(os.listdir(out_root))
        assert n_dirs == n_cli, (n_dirs, n_cli)
        return {"metric": "complete on-disk triples/s (graph CSV + 304x304 image PNG + 1216x1216 label PNG per sample)", "value": len(seeds) / dt,
                "unit": "samples/s", "samples": len(seeds), "seconds": dt, "generate_seconds": t_gen, "write_seconds": dt - t_gen,
                "writer_threads": threads or default_threads(), "bytes_written": nbytes, "where": out_root.rsplit("/", 1)[0],
                "cli_pipelined": {"value": n_cli / dt_cli, "unit": "samples/s", "samples": n_cli, "seconds": dt_cli,
                                  "command": f"generate_vessel_graph.py --num_samples {n_cli} --labels (defaults: --batch 512 --inflight 2, 16 writer threads)",
                                  "note": "the drop-in CLI end to end, simulator set-up of its generator threads included (process start excluded): config.yml + CSV + image PNG + label PNG per sample"},
                "note": "one batch, nothing overlapped: simulate + rasterise on the GPU, then native CSV formatting (byte-identical to numpy's "
                        "str(ndarray) / repr(float), tests/test_fileio.py) and PNG encoding on host threads"}
    finally:
        writer.close()
        shutil.rmtree(out_root, ignore_errors=True)


def train_cli_leg(n_graphs=128, epochs=4, extra_args=()):
    """BASELINE configs[2] through the reference's entry point: `train.py --config_file configs/config_ves_seg-S.yml` on freshly
    generated full-length graphs (the reference's 500 provided pairs are not on the GPU box): graph CSV -> loader (parse, two
    rasterisations per sample, augmentation) -> DynUNet-S step at 1216^2. Reports the last epoch's images per second."""
    import yaml
    import generate_vessel_graph
    import train as train_cli
    from octa_autosegmentation_amd.utils import configs
    tmp = tempfile.mkdtemp(prefix="octa_bench_train_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    old_stdout = sys.stdout
    try:
        sys.stdout = sys.stderr                          # the CLIs print progress; stdout carries the JSON line only
        generate_vessel_graph.main(["--config_file", configs.GENERATOR_CONFIG, "--num_samples", str(n_graphs), "--seed", "1",
                                    "--output.directory", os.path.join(tmp, "graphs")])
        cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "config_ves_seg-S.yml")))
        cfg.pop("Validation"); cfg.pop("Test")
        p = os.path.join(tmp, "cfg.yml")
        with open(p, "w") as f:
            yaml.safe_dump(cfg, f)
        csvs = os.path.join(tmp, "graphs", "**", "*.csv")
        train_cli.main(["--config_file", p, "--Train.data.image.files", csvs, "--Train.data.label.files", csvs, "--Train.epochs", str(epochs),
                        "--Train.epochs_decay", "0", "--General.seed", "3", "--Output.save_dir", os.path.join(tmp, "results")] + list(extra_args))
        rates = list(train_cli.LAST_RUN["imgs_per_s"])
        return {"metric": "train.py imgs/s with configs/config_ves_seg-S.yml (graph CSVs -> device-side loader -> DynUNet-S step @1216^2, bf16, B=4)",
                "value": rates[-1], "unit": "imgs/s", "epochs": epochs, "graphs": n_graphs, "imgs_per_s_per_epoch": rates,
                "note": "epoch 1 includes parsing the CSV files (cached on the device afterwards) and first-call overheads"}
    finally:
        sys.stdout = old_stdout
        shutil.rmtree(tmp, ignore_errors=True)


def launcher_selftest(args):
    """The N-rank skeleton of main() with a stub body, on gloo (no GPU): process group from the launcher's environment, the rank count by
    all-reduce, a barrier-bracketed timed region, MAX of the wall time over ranks, per-rank values by all-gather, ONE JSON line from rank 0."""
    import torch
    import torch.distributed as dist
    from octa_autosegmentation_amd.utils import sharding
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        dist.init_process_group(backend="gloo")
    ones = torch.ones(1)
    if world > 1:
        dist.all_reduce(ones)
        dist.barrier()
    t0 = time.time()
    seeds = np.concatenate([sharding.rank_seeds(rank, i, args.batch) for i in range(args.steps)])      # the stub step: this rank's seed blocks
    time.sleep(0.01 * (rank + 1))
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    per_rank = None
    if world > 1:
        t_all = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(t_all, torch.tensor([dt], dtype=torch.float64))
        per_rank = [args.batch * args.steps / float(t.item()) for t in t_all]
        s_all = [torch.zeros(2, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(s_all, torch.tensor([int(seeds.min()), int(seeds.max())]))
    dt = sharding.max_over_ranks(dt, dist if world > 1 else None)
    if rank == 0:
        print(json.dumps({"metric": "launcher selftest", "value": world * args.batch * args.steps / dt, "unit": "samples/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "rccl_ranks": int(ones.item()), "per_rank_value": per_rank, "self_launched": os.environ.get("OCTA_SELF_LAUNCHED") == "1",
                          "seed_ranges": [[int(a), int(b)] for a, b in (t.tolist() for t in s_all)] if world > 1 else None}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--inflight", type=int, default=2, help="launches in flight per GPU (each on its own HIP stream); with the default --serial-sim only ONE "
                    "persistent kernel runs at a time and the other launch is being rasterised meanwhile")
    ap.add_argument("--group", type=int, default=4, help="steps (128-sample batches) simulated by ONE launch of the persistent kernel. Round 3: a "
                    "launch of 4 x 128 samples fills the GPU's 512 workgroup slots (two samples per CU) and the rasteriser then has the whole GPU; "
                    "measured on one box: group / inflight 4/1 681, 4/2 644, 2/2 620, 1/4 616, 2/3 567 samples/s (rounds 1-2: 1 step per launch, 4 in flight)")
    ap.add_argument("--no-serial-sim", dest="serial_sim", action="store_false", help="with --inflight > 1: let the persistent kernels of several launches share the GPU "
                    "(default: one persistent kernel at a time, a launch's rasterisation overlaps the NEXT launch's kernel and fills its tail; measured on one box, "
                    "group 4: in flight 1: 725, 2 serial: 744, 3 serial: 739, 2 concurrent: 644 samples/s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="do not run the two rocprofv3 counter passes (roofline.traffic then comes from the committed file)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-train", action="store_true", help="skip the secondary U-Net training measurement")
    ap.add_argument("--train-batch", type=int, default=4)
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the on-the-fly generation + training measurement")
    ap.add_argument("--no-files", action="store_true", help="skip the on-disk triples leg")
    ap.add_argument("--cpu-all-cores", action="store_true", help="also time the CPU baseline with one worker per visible core (BASELINE.md 3: 'all cores'; "
                    "230 s on the 256-thread host of the GPU box, hence not in the default run)")
    ap.add_argument("--no-long", dest="long", action="store_false", help="skip BASELINE configs[4] at its stated size: a 10 000-sample on-the-fly epoch (2 500 training "
                    "steps of 4 over all ranks; about a minute on one MI355X). Round 5: part of the DEFAULT run, so that the driver's line carries end_to_end_10k_epoch")
    ap.add_argument("--long", dest="long", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--launcher-selftest", action="store_true", help=argparse.SUPPRESS)      # tests/test_bench_launcher.py: the launcher + one-line contract on gloo ranks, no GPU
    ap.set_defaults(long=True)
    args = ap.parse_args()
    if args.pmc_child:
        pmc_child(args.batch)
        return
    # `python bench.py --gpus N` starts its own N ranks (one per GPU, RCCL over 127.0.0.1); under torch.distributed.run nothing is re-executed
    from octa_autosegmentation_amd.utils import launch as _launch
    try:
        if _launch.needs_self_launch(args.gpus):
            sys.exit(_launch.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus, need_devices=not args.launcher_selftest))
        _launch.check_world(args.gpus)
    except _launch.LaunchError as e:
        sys.exit(f"bench.py: {e}")
    if args.launcher_selftest:
        launcher_selftest(args)
        return

    # stdout carries exactly ONE line, the JSON record: libraries that print banners from C (RCCL's version block at communicator
    # set-up, MIOpen notes) and the CLIs this bench drives in-process write to file descriptor 1 behind python's back -- from here on
    # descriptor 1 IS stderr, and the record goes to the saved descriptor of the real stdout at the very end
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a ROCm GPU (no CPU fallback)"
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    dist = None
    if world > 1:
        import torch.distributed as dist
        # one process per GPU; "nccl" is RCCL on ROCm. The device is bound first so that the communicator
        # and its barrier live on this rank's GPU.
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    dev = torch.device("cuda", torch.cuda.current_device())
    rccl_ranks = 1
    host = None
    if world > 1:
        from octa_autosegmentation_amd.utils import sharding as _sh
        host = _sh.apply_host_budget(generator_threads=max(1, args.inflight))       # per-rank share of the host: service-thread spin, BLAS threads, CPU set
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                                                        # every rank that answers adds one: the number of RCCL ranks really in the job
        rccl_ranks = int(ones.item())
    global N_CUS
    N_CUS = int(torch.cuda.get_device_properties(dev).multi_processor_count)

    from concurrent.futures import ThreadPoolExecutor
    from octa_autosegmentation_amd import _native, pipeline
    from octa_autosegmentation_amd.utils import sharding
    cfg = load_config()
    B = args.batch
    n_fly = max(1, args.inflight)
    sim_conc = 1 if (args.serial_sim or n_fly == 1) else n_fly          # persistent kernels on the GPU at a time
    G = max(1, args.group)
    # One launch of the persistent kernel simulates G steps (G x B samples; at most one workgroup per CU, workgroups pull samples
    # from a work queue); n_fly such launches are in flight, each with its own simulator state and HIP stream, so that one launch's
    # host side (seeding, edge export, rasterisation) and its tail overlap the others' kernels: n_fly x G x B workgroups' worth of
    # samples for 256 CUs.
    sizes = sorted({G} | ({args.warmup % G} if args.warmup % G else set()) | ({args.steps % G} if args.steps % G else set()))
    gens = {g: [pipeline.TripleGenerator(cfg, B * g) for _ in range(n_fly)] for g in sizes}
    streams = [torch.cuda.Stream() for _ in range(n_fly)]
    if args.serial_sim and n_fly > 1:
        gate = pipeline.SimGate()
        for gl in gens.values():
            for g_ in gl:
                g_.sim_gate = gate

    def launch(slot, first_step, nsteps, wait=True):
        seeds = np.concatenate([sharding.rank_seeds(rank, i, B) for i in range(first_step, first_step + nsteps)])
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[slot]):
            out = gens[nsteps][slot].generate(seeds)
            if wait:
                finish(slot, out)
        return out

    def finish(slot, out):
        ev = out.get("done_event")
        if ev is not None:
            ev.synchronize()
        else:
            streams[slot].synchronize()
        out["wall"]["t_done"] = time.time()

    def chain(slot, groups):
        # A slot asks for its NEXT launch before it waits for its previous result (round 6): the thread then stands at the gate while the
        # other slot's kernel runs, so every finished launch finds a successor waiting and its rasterisation is ordered behind that launch
        # (and runs in its tail) instead of going ahead alone with the next launch ordered behind ALL of it. Stream order keeps it safe: a
        # generator's launches share one stream, its rasterisations another.
        outs = []
        for f0, n in groups:
            same_gen = bool(outs) and outs[-1]["steps"] == n
            if outs and not same_gen:
                finish(slot, outs[-1])
            o = launch(slot, f0, n, wait=False)
            o["steps"] = n
            if outs and same_gen:
                finish(slot, outs[-1])
            outs.append(o)
        if outs:
            finish(slot, outs[-1])
        return outs

    pool = ThreadPoolExecutor(max_workers=n_fly)

    # One-time set-up of every generator the timed region uses (not a step: nothing is simulated). With W = 5 warm-up steps and 4 steps per
    # launch only ONE slot's full-size generator runs before the clock starts; the other slot's first launch would grow its rasteriser scratch
    # (2.7 GB of polygon sides, 1.4 GB of row lists: hipMalloc + an implicit device synchronisation each), its output pool and load the render
    # kernels inside the timed region. A synthetic edge list of a launch's size goes through the image and the label rasterisation of each.
    for g_ in gens[G]:
        with torch.cuda.stream(streams[gens[G].index(g_)]):
            g_.prime()
    torch.cuda.synchronize()

    def run_steps(first, count):
        # slot-affine: launch j runs on slot j % n_fly, one launch per slot at a time
        groups = [(first + k, min(G, count - k)) for k in range(0, count, G)]
        chains = [[g for j, g in enumerate(groups) if j % n_fly == s] for s in range(n_fly)]
        futs = [pool.submit(chain, s, ch) for s, ch in enumerate(chains)]
        outs = []
        for f in futs:
            outs.extend(f.result())
        return outs

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        run_steps(0, args.warmup)
    barrier()
    import gc
    gc.collect()
    gc.disable()                       # no collector pause between two launches of the timed region (it holds every thread: 5 launches are timed)
    try:
        t0 = time.time()
        outs = run_steps(args.warmup, args.steps)
        barrier()
        dt = time.time() - t0
    finally:
        gc.enable()
    # where a slot's time goes (host clock, mean over the timed launches): simulator call (host init + kernel + download + BFS export),
    # render enqueue, wait for the render kernels; `kernel` is the device time of the persistent kernel inside the simulator call
    slot_cycle = None
    if outs and "wall" in outs[0]:
        w = [o["wall"] for o in outs]
        slot_cycle = {"sim_call_ms": 1e3 * float(np.mean([x["sim_run_s"] for x in w])),
                      "kernel_ms": float(np.mean([o["result"].timing["kernel_b_ms"] for o in outs])),
                      "render_enqueue_ms": 1e3 * float(np.mean([x["render_enqueue_s"] for x in w])),
                      "render_wait_ms": 1e3 * float(np.mean([x["t_done"] - x["t_start"] - x["sim_run_s"] - x["render_enqueue_s"] for x in w])),
                      "launch_to_done_ms": 1e3 * float(np.mean([x["t_done"] - x["t_start"] for x in w]))}
        print(f"[bench] slot cycle: {slot_cycle}", file=sys.stderr)
        # launch by launch (stderr only): kernel time, whether the rasterisation was ordered behind a successor and what its gate kernel saw
        for o in sorted(outs, key=lambda o_: o_["wall"]["t_start"]):
            g_ = o["wall"].get("gate")
            print(f"[bench]   launch ticket {o['wall'].get('ticket')}: start {o['wall']['t_start'] - t0:7.3f} s  kernel {o['result'].timing['kernel_b_ms']:6.1f} ms  sim call {1e3 * o['wall']['sim_run_s']:6.1f}  "
                  f"successor {o['wall'].get('ordered_behind_successor')}  gate {g_.cpu().tolist() if g_ is not None else None}", file=sys.stderr)
            if o["wall"].get("host"):
                print(f"[bench]     host: gate -> run {o['wall']['host']['gate_to_run_ms']:.1f} ms, run {o['wall']['host']['run_ms']:.1f} (loop {o['result'].timing['loop_wall_ms']:.1f}, kernel {o['result'].timing['kernel_b_ms']:.1f}), "
                      f"plan inside the gate {o['wall']['host']['plan_in_gate_ms']:.1f}; waited for the gate {1e3 * (o['wall']['t_start'] - o['wall']['t_request']):.1f}", file=sys.stderr)
            sp_ = o["result"].spans.astype(np.float64) / 1e5                 # ms on the device's 100 MHz clock
            b0 = sp_[:, 0].min()
            print("[bench]     sample starts after the first (ms): " + " ".join(f"p{q}={np.percentile(sp_[:, 0] - b0, q):.1f}" for q in (25, 50, 60, 75, 90, 100))
                  + "   ends: " + " ".join(f"p{q}={np.percentile(sp_[:, 1] - b0, q):.1f}" for q in (0, 25, 50, 75, 100)), file=sys.stderr)
    ka = kb = 0.0
    la = lb = 0
    bif_ms = 0.0
    relaunches = 0
    # does the host's bifurcation service ever gate a workgroup? Device side: the "mailbox" phase timer (octa_sim_stats slot 13, 100 MHz ticks) is
    # the time a sample's workgroup spent waiting for its answers over the whole run; host side: the service loop's record
    from octa_autosegmentation_amd.vessel_graph_generation import greenhouse as _gh
    svc = [o["result"].service for o in outs]
    wait_ms = float(np.mean([o["result"].stats[:, 13].mean() for o in outs])) * 1e-5
    mailbox = {"service": _gh.bifurcation_service_kind()[0], "why_not_native": _gh.bifurcation_service_kind()[1],
               "tickets_per_launch": float(np.mean([v["tickets"] for v in svc])), "parked_workgroups": int(sum(v["parked"] for v in svc)),
               "relaunches": int(sum(v["relaunches"] for v in svc)), "longest_callback_ms": float(max(v["max_callback_ms"] for v in svc)),
               "longest_service_absence_ms": float(max(v["max_absence_ms"] for v in svc)),
               "device_wait_ms_per_sample": wait_ms, "per_sample_device_ms": None,
               "note": "host_bifurcation_callback_ms_per_step is host LAPACK time summed over a step's samples; it runs beside the kernel. What a "
                       "workgroup pays is device_wait_ms_per_sample (its ~55 round trips of the whole run together, the other workgroup of the CU keeps "
                       "working meanwhile); a stall would show as parked_workgroups / relaunches > 0 (a workgroup that waited 20 ms leaves the kernel)"}
    for out in outs:
        r = out["result"]
        tm = r.timing
        ka += tm["kernel_a_ms"]; kb += tm["kernel_b_ms"]; la += tm["launches_a"]; lb += tm["launches_b"]
        bif_ms += tm["host_bif_ms"]
        relaunches += r.service["relaunches"]
        assert int(r.stats[:, 0].max()) == 0, "simulator reported error bits"
    # device milliseconds one sample spends in its ten phases (100 MHz timers of thread 0: octa_sim_stats slots 0-9, mailbox waits
    # included; slots 10-15 are sub-timers of those phases) while n_fly launches share the GPU
    phase_ms = lambda st: float(st[:, 8:18].sum(axis=1).mean()) * 1e-5
    sample_ms_loaded = float(np.mean([phase_ms(o["result"].stats) for o in outs]))
    mailbox["per_sample_device_ms"] = sample_ms_loaded
    # share of the CU time the simulator's workgroups held during the timed steps: per-sample spans on the device's common 100 MHz
    # clock (octa_sim_spans), clipped to a window inside the steady state (between the quartiles of the first-taken / last-left times)
    sp = np.concatenate([o["result"].spans for o in outs]).astype(np.float64) / 1e8
    w0, w1 = np.percentile(sp[:, 0], 25), np.percentile(sp[:, 1], 75)
    cu_time = None
    if w1 > w0:
        held = float(np.clip(np.minimum(sp[:, 1], w1) - np.maximum(sp[:, 0], w0), 0, None).sum())
        geo_ = np.zeros(4, np.int32)
        _native.check(_native.lib().octa_sim_geometry(N_CUS, geo_.ctypes.data), "octa_sim_geometry")
        cu_time = {"simulator_share": held / (N_CUS * int(geo_[1]) * (w1 - w0)), "workgroup_slots": N_CUS * int(geo_[1]), "span_ms_per_sample": float((sp[:, 1] - sp[:, 0]).mean() * 1e3),
                   "window_s": float(w1 - w0),
                   "note": "slot-seconds held by simulator workgroups / (CUs x workgroup slots per CU x window); the remainder is the rasteriser's "
                           "kernels (they need CUs free of simulator workgroups) and dispatch gaps"}

    # ---- rasteriser alone (other slots idle): HIP events on the stream the kernels go to
    raster = None
    sample_ms, solo_launch_ms = sample_ms_loaded, None
    if rank == 0:
        g0 = pipeline.TripleGenerator(cfg, B)                            # ONE 128-sample batch: every sample alone on its CU
        g0.time_render = True
        best = None
        solo = []
        for rep in range(3):
            with torch.cuda.stream(streams[0]):
                o = g0.generate(sharding.rank_seeds(rank, 900 + rep, B))
                streams[0].synchronize()
            ms = pipeline.TripleGenerator.render_ms(o)
            best = ms if best is None else {k: min(best[k], ms[k]) for k in ms}
            solo.append((phase_ms(o["result"].stats), o["result"].timing["kernel_b_ms"]))
        sample_ms = float(np.mean([a for a, _ in solo]))            # one launch with the GPU to itself
        solo_launch_ms = float(np.mean([b for _, b in solo]))
        g0.close()
        lab, img = best["label_raster_ms"], best["image_raster_ms"]
        raster = {"kernel": "octa_rasterize_2d launch sequence (raster_meta / scan / tess / render)", "bound": "hbm",
                  "label_1216": {"ms_per_batch": lab, "achieved": RASTER_BYTES_1216 * B / (lab * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": RASTER_BYTES_1216 * B / (lab * 1e-3) / 1e9 / HBM_PEAK_GBS},
                  "image_304_x2": {"ms_per_batch": img, "achieved": RASTER_BYTES_304 * 2 * B / (img * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": RASTER_BYTES_304 * 2 * B / (img * 1e-3) / 1e9 / HBM_PEAK_GBS},
                  "read_back_ms": best["read_back_ms"], "dither_ms": best["dither_ms"],
                  "note": "HBM-bound on paper (2.2 MB per label), ALU-bound in the per-pixel coverage fold in practice (DESIGN.md 4.2)"}

    files_info = None
    if rank == 0 and not args.no_files:
        gf = pipeline.TripleGenerator(cfg, B)
        files_info = files_leg(gf, streams[0], sharding.rank_seeds(rank, 950, B))
        gf.close()

    # secondary metric of BASELINE.json: DynUNet-S training images/s at 1x1216x1216, bf16, on the MFMA convolution
    # path; reported so the gap to the 200 imgs/s target is tracked, it is NOT part of `value`.
    train_info = None
    for lst in gens.values():
        for g_ in lst:
            g_.close()
    gens = {}
    if not args.no_train:
        torch.cuda.empty_cache()
        train_info = unet_train_bench(dev, args.train_batch, dist, world, with_allreduce=True)
        # B in {4, 8, 16} at 1 x 1216 x 1216 (12.9 / 25.7 GiB of activations)
        train_info["other_batch_sizes"] = {}
        for b_ in (8, 16):
            torch.cuda.empty_cache()
            r_ = unet_train_bench(dev, b_, dist, world, steps=8, warmup=2)
            train_info["other_batch_sizes"][f"B{b_}"] = {"value": r_["value"], "unit": "imgs/s", "ms_per_step": r_["ms_per_step"], "batch_per_gpu": b_,
                                                         "mfma_frac": r_["roofline"]["mfma"]["frac"]}
    cli_info = None
    if not args.no_train and not args.no_end_to_end and world == 1:
        torch.cuda.empty_cache()
        cli_info = train_cli_leg()
    elif world > 1:
        cli_info = {"value": None, "skipped": f"world size {world}: this leg drives train.py's main() in-process on graphs it generates under /dev/shm of ONE rank; the "
                                              "data-parallel trainer is measured by unet_train / end_to_end_* (every rank) and train.py under torch.distributed.run by "
                                              "tests/test_training_cli.py (two gloo ranks)"}
    # BASELINE.json configs[4]: on-the-fly simulation + rasterisation + GPU augmentation feeding the same training step
    e2e_info = e2e_gan_info = None
    if not args.no_train and not args.no_end_to_end:
        import train_synthetic
        torch.cuda.empty_cache()
        # generator batches of 256 samples (round 3: a sample holds half a CU, so 256 workgroups = the CU time 128 held before; with 128
        # the single generator launch in flight produced fewer samples per second than the training step consumes): warm-up = one
        # generator batch (the queue-filling transient), then three batches timed
        # round 5: generator batches of 512 samples = every workgroup slot of the GPU (two per CU), where the persistent kernel is cheapest per sample
        # (same box: 174.6 imgs/s with 256-sample batches, 182.4 with 512; making the trainer step aside for the generator's launch -- an exclusive
        # burst -- measured the same 182.0 and was not kept); the timed window is two generator batches (2 x 128 steps)
        e2e_info = train_synthetic.run(steps=256, batch=args.train_batch, gen_batch=512, seed0=500000, log=False, warmup=128)
        # configs[4] proper: the same stream feeding the joint GAN contrast-adaptation + segmentation step (G, D at 304^2, S at 1216^2)
        torch.cuda.empty_cache()
        e2e_gan_info = train_synthetic.run(steps=128, batch=args.train_batch, gen_batch=512, seed0=600000, log=False, gan=True, warmup=64)

    per_rank = None
    if dist is not None:
        t_all = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(t_all, torch.tensor([dt], dtype=torch.float64, device=dev))
        per_rank = [B * args.steps / float(t.item()) for t in t_all]
    dt = sharding.max_over_ranks(dt, dist, dev)
    voxel_info = gan_net_info = long_info = None
    if rank == 0 and not args.no_train:
        torch.cuda.empty_cache()
        voxel_info = voxel_leg(cfg, dev)
        gan_net_info = gan_networks_leg(dev)
    if args.long and not args.no_train:
        import train_synthetic
        torch.cuda.empty_cache()
        long_info = train_synthetic.run(steps=2500 // max(world, 1) + 64, batch=args.train_batch, gen_batch=512, seed0=700000, log=False, warmup=64)
        if long_info is not None:
            long_info["note"] = "BASELINE configs[4] at its stated size: 10 000 samples per epoch over all ranks (2 500 steps of 4 + warm-up)"

    if dist is not None:
        # every collective of the run is behind us: the group ends HERE, so that rank 0's host-side legs below (the CPU baseline on the host's
        # cores, the counter passes) hold no other rank inside a collective -- the other ranks simply leave
        dist.barrier()
        dist.destroy_process_group()
        dist = None
    if rank == 0:
        value = world * B * args.steps / dt
        per_gpu = value / world
        # dominant kernel: the persistent simulator kernel (one launch per batch runs all iterations of every
        # sample; OCTA_SIM_LOCKSTEP=1 selects the two-launches-per-iteration form, then launch A or B)
        if la == 0:
            dom_ms, dom_n, dom_name = kb, lb, "sim_persistent_kernel"
            bytes_per_launch = ALGO_BYTES_PER_SAMPLE * B * args.steps / max(lb, 1)
            note = (f"one launch = 250 dependent growth iterations of {G} x {B} independent samples (work queue, two workgroups = two samples per CU); "
                    "dependency/latency-bound (ordered passes, pow chains), not HBM-bound: see serial_depth")
        else:
            dom_ms, dom_n, dom_name = (kb, lb, "sim_iter_b_kernel") if kb >= ka else (ka, la, "sim_iter_a_kernel")
            n_iter = max(lb // max(args.steps, 1), 1)
            bytes_per_launch = ALGO_BYTES_PER_SAMPLE / (2.0 * n_iter) * B
            note = "lock-step form: 250 dependent iterations x 2 launches; latency-bound, see serial_depth"
        launch_ms = dom_ms / max(dom_n, 1)
        achieved = bytes_per_launch / (launch_ms * 1e-3) / 1e9
        traffic = None
        if world == 1 and not args.no_pmc and la == 0:
            per_launch = max(1, args.steps * B // max(dom_n, 1))            # samples one launch of this run simulates
            raster_extra = {}
            traffic, traffic_src = pmc_traffic_live(dom_name, per_launch, raster_extra)
            if traffic is not None and raster is not None and raster_extra:
                per_kernel = {k: {"launches": v["launches"], "fetch_bytes": 2.0 * v["FETCH_SIZE"] * 1024.0, "write_bytes": v["WRITE_SIZE"] * 1024.0,
                                  "bytes": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0} for k, v in sorted(raster_extra.items())}
                total = sum(v["bytes"] for v in per_kernel.values())
                algo = per_launch * (RASTER_BYTES_1216 + 2 * RASTER_BYTES_304)
                raster["traffic"] = {"bytes": total, "samples": per_launch, "algorithmic_bytes": algo, "ratio": total / algo,
                                     "write_bytes": sum(v["write_bytes"] for v in per_kernel.values()),
                                     "output_bytes": per_launch * (1216 * 1216 * 2 + 3 * 304 * 304),
                                     "per_kernel": per_kernel,
                                     "unit": "bytes per generate() of `samples` triples: both 304^2 rasterisations + max, CSV read-back emulation, the 1216^2 label "
                                             "rasterisation and the dither together (2 x FETCH_SIZE + WRITE_SIZE of the same two counter passes as roofline.traffic); "
                                             "output_bytes = grey label + binarised label + two 304^2 rasters + their maximum"}
            if traffic is None:
                print(f"[bench] live counter passes unavailable ({traffic_src}); using the committed summary", file=sys.stderr)
        if traffic is None:
            samples_per_launch = args.steps * B / max(dom_n, 1)
            traffic, traffic_src, exact = pmc_traffic_per_launch(dom_name, int(min(samples_per_launch, N_CUS * 2)) * 256)
            if traffic is not None:
                if not exact:      # rows of 128-sample launches only: scale (an UNDER-estimate: a full GPU misses L2 more)
                    traffic = traffic * samples_per_launch / 128.0
                traffic_src = f"NOT measured in this run: committed file {traffic_src}" + ("" if exact else ", scaled from 128-sample launches")
        geo = np.zeros(4, np.int32)
        _native.check(_native.lib().octa_sim_geometry(N_CUS, geo.ctypes.data), "octa_sim_geometry")
        wg_per_cu = int(geo[1])
        # every CU holds wg_per_cu samples at a time; with all of them resident a sample takes sample_ms_loaded
        bound_samples_s = N_CUS * wg_per_cu / (sample_ms_loaded * 1e-3)
        line = {
            "metric": "synthetic OCTA samples/sec (graph + 304x304 image + 1216x1216 label triples)",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[1]: {B}-sample vessel-graph batch (docker/vessel_graph_gen_docker_config.yml, "
                                   f"I=100+150, N=2000) + tree2img rasterise 304x304 image and 1216x1216 label",
                       "batch_per_gpu": B, "steps_per_launch": G, "launches_in_flight": n_fly, "persistent_kernels_at_a_time": sim_conc, "parallelism": f"sample-sharded x{world}, no collective",
                       "set_up": "every full-size generator primed once before the clock starts (TripleGenerator.prime: a synthetic edge list through its rasteriser: scratch growth and kernel "
                                 "loading; nothing simulated); Python's cyclic garbage collector is off inside the timed region (gc.disable: a collection holds every thread)",
                       "ordering": "a launch's rasterisation is ordered behind the NEXT launch on the device (csrc/order.hip gate kernel on the rasteriser's stream; no host polling or sleeps); "
                                   "a slot asks for its next launch before it waits for its previous result, so every finished launch finds its successor at the gate"},
            "parity": "graph CSV text bit-exact with the REFERENCE (imported and run in the build container) on 8 short + 2 full-length fixture "
                      "runs and on 64 further full-length seeds (tests/golden/sim_wide_golden.npz: SHA-256 of the CSV text of seeds 1000-1063; the "
                      "GPU reproduces all 64, tests/test_sim_gpu.py); label / image pixels bit-exact on the reference's fixtures. The oracle follows "
                      "glibc's acos, numpy on the AVX-512 build host its own SIMD arccos: the oracle's edge lists differ from the reference's in 0-241 "
                      "of ~92 000 doubles per full-length sample (last bit), never in a printed digit on those 64 + 10 runs; GPU and oracle agree in "
                      "EVERY double (radii: glibc pow restated; node positions: glibc acos / sin / cos restated, csrc/glibc_trig.h) on the validated "
                      "full-length seeds (profiles/r02_validate_final.log, profiles/r03_validate.log)",
            "rccl_ranks": rccl_ranks, "per_rank_value": per_rank, "host_budget": host,
            "self_launched": os.environ.get("OCTA_SELF_LAUNCHED") == "1",
            "value_with_csv": (files_info or {}).get("cli_pipelined", {}).get("value") if files_info else None,
            "value_with_csv_note": "complete ON-DISK triples per second (config.yml + graph CSV + 304x304 image PNG + 1216x1216 label PNG per sample) through the "
                                   "drop-in CLI generate_vessel_graph.py --labels: the triple as SURVEY.md 8d defines it; `value` counts triples complete in HBM",
            "roofline": {"bound": "hbm", "binds": "latency: the serial dependency depth of one sample's 250 iterations (see serial_depth), not HBM bandwidth and not the matrix cores",
                         "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_unit": f"bytes per launch (2 x FETCH_SIZE + WRITE_SIZE; {traffic_src})",
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": launch_ms, "launches": dom_n, "note": note,
                         "serial_depth": {"per_sample_device_ms": sample_ms, "per_sample_device_ms_all_slots_taken": sample_ms_loaded,
                                          "solo_launch_ms": solo_launch_ms, "cus": N_CUS, "workgroups_per_cu": wg_per_cu,
                                          "threads_per_workgroup": int(geo[0]), "lds_bytes_per_workgroup": int(geo[2]),
                                          "bound_samples_per_s": bound_samples_s, "frac_of_bound": per_gpu / bound_samples_s,
                                          "note": f"a sample occupies one of the {wg_per_cu} workgroup slots of a CU ({int(geo[2]) // 1024} KiB of LDS, "
                                                  f"{int(geo[0])} threads at 256 registers) for per_sample_device_ms when its launch has the GPU to itself "
                                                  "(one sample per CU: the solo leg runs 128 samples) and for ..._all_slots_taken in the timed launches (the co-resident "
                                                  "sample shares the CU's issue slots, LDS and L1); CUs x slots / that time is what the simulator could "
                                                  "reach alone on the GPU; this dependency chain, not HBM, is the binding limit"},
                         "rasteriser": raster, "voxeliser": voxel_info, "gan_networks": gan_net_info},
            "kernel_ms_per_launch": {dom_name: launch_ms, "launches_per_step": dom_n / max(args.steps, 1),
                                     "cu_occupancy_weighted_ms_per_step": launch_ms * (dom_n / max(args.steps, 1)) / sim_conc,
                                     "note": f"a launch covers {G} steps; {n_fly} launches are in flight and {sim_conc} persistent kernel(s) run at a time (the other "
                                             "launch is being rasterised meanwhile); the weighted figure is the launch duration per step divided by the persistent "
                                             "kernels running at a time"},
            "cu_time": cu_time,
            "slot_cycle": slot_cycle,
            "host_bifurcation_callback_ms_per_step": bif_ms / args.steps,
            "mailbox_relaunches": relaunches,
            "mailbox": mailbox,
        }
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, all_cores=args.cpu_all_cores)
            if world > 1:
                line["cpu_baseline"]["note_ranks"] = f"timed by rank 0 after the process group ended (the other {world - 1} ranks had left: the host's cores were free)"
            if not args.no_train:
                line["cpu_baseline"]["unet_train_step"] = cpu_unet_step()
        line["files"] = files_info
        line["unet_train"] = train_info
        line["train_cli"] = cli_info
        line["end_to_end_train"] = e2e_info
        line["end_to_end_gan_seg_train"] = e2e_gan_info
        line["end_to_end_10k_epoch"] = long_info
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(line) + "\n").encode())


if __name__ == "__main__":
    main()
