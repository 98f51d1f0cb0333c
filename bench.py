"""bench.py -- headline benchmark: synthetic OCTA triples per second on MI355X.

A step = one pass of the hot path over one batch of seeded samples with the reference's own
generator config (docker/vessel_graph_gen_docker_config.yml):
space-colonisation simulation of B vessel graphs (HIP), 304x304 arterial/venous rasterisation and
max-combine, 1216x1216 label rasterisation + Floyd-Steinberg binarisation. Outputs stay in memory
(edge arrays on the host, images/labels in HBM); writing CSV/PNG files is not part of the step.
Workload = BASELINE.json configs[1] (128-sample batch, rasterise to 1216x1216).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
Multi-GPU (driver-launched with torch.distributed.run): samples are independent, every rank
generates its own batch with its own seeds; no data-path collective (weak scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# one HIP hardware queue per step in flight (the runtime's default of 4 would make streams share queues)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_SAMPLE = 84e6       # point traffic of one full-length sample (all iterations)
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: 8.0 TB/s spec


def pmc_traffic_per_launch(kernel_name):
    """HBM-side bytes per launch of `kernel_name` from the committed rocprofv3 counter passes
    (profiles/r01_bench_pmc_summary.csv: FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes of this
    bench, values in KB). Correction per MI355X_MICROARCH.md (HBM): FETCH_SIZE counts 128-B requests as 64 B on
    gfx950, so it is doubled; WRITE_SIZE is taken as reported. None if the file or the kernel is missing."""
    path = os.path.join(ROOT, "profiles", "r01_bench_pmc_summary.csv")
    if not os.path.exists(path):
        return None
    import csv
    kb = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel_name in r["kernel"]:
                kb[r["counter"]] = float(r["avg_KB_per_launch"])
    if "FETCH_SIZE" not in kb or "WRITE_SIZE" not in kb:
        return None
    return (2.0 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024.0


def load_config():
    from octa_autosegmentation_amd.utils import configs
    return configs.load_generator_config()


def cpu_baseline(cfg, budget_s=30.0):
    """The oracle (CPU restatement of the reference) on host cores: full-length samples, one core."""
    from oracle import octa_oracle, sim_oracle
    from octa_autosegmentation_amd import graph_io
    t0 = time.time()
    n = 0
    while True:
        e, info = sim_oracle.simulate(cfg, 900 + n)
        na = info["n_art_edges"]
        np.maximum(octa_oracle.rasterize(e[:na], [304, 304]), octa_oracle.rasterize(e[na:], [304, 304]))
        octa_oracle.fs_dither(octa_oracle.rasterize(graph_io.edges_as_read_back(e), [1216, 1216]))
        n += 1
        if time.time() - t0 > budget_s * 0.6 or n >= 3:
            break
    dt = time.time() - t0
    return {"value": n / dt, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"{n} full-length samples (I=100+150, N=2000) incl. 304x304 image + 1216x1216 label, oracle/ C++ on one core"}


def unet_train_bench(dev, batch, dist, world, steps=20, warmup=4):
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
    return {"metric": "DynUNet-S training imgs/s @1x1216x1216", "value": ips, "unit": "imgs/s", "dtype": "bf16",
            "batch_per_gpu": batch, "ms_per_step": dt / steps * 1e3, "tflops": 2.0 * ips,
            "frac_of_bf16_dense_peak": 2.0 * ips / 2500.0,
            "implementation": "channels-last bf16 on hand-written HIP kernels: MFMA 3x3 conv forward / data gradient / weight "
                              "gradient (csrc/conv.hip; skip-connection gradients in the data-gradient epilogue), NHWC InstanceNorm+LeakyReLU "
                              "(csrc/norm.hip; the last one fused with the 1x1 output convolution), one-launch weight packing; hipBLASLt "
                              "only for the 1x1 transposed convolution at the bottleneck; flat RCCL gradient all-reduce"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--inflight", type=int, default=4, help="steps in flight per GPU (each on its own HIP stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="skip the secondary U-Net training measurement")
    ap.add_argument("--train-batch", type=int, default=4)
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the on-the-fly generation + training measurement")
    args = ap.parse_args()

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

    from concurrent.futures import ThreadPoolExecutor
    from octa_autosegmentation_amd import pipeline
    from octa_autosegmentation_amd.utils import sharding
    cfg = load_config()
    B = args.batch
    n_fly = max(1, args.inflight)
    # n_fly independent 128-sample steps are kept in flight, each with its own simulator state and HIP
    # stream: while the host serves one step's (rare) LAPACK bifurcation requests the GPU advances the other
    gens = [pipeline.TripleGenerator(cfg, B) for _ in range(n_fly)]
    streams = [torch.cuda.Stream() for _ in range(n_fly)]

    def step(i):
        slot = i % n_fly
        seeds = sharding.rank_seeds(rank, i, B)
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[slot]):
            out = gens[slot].generate(seeds)
            streams[slot].synchronize()
        return out

    pool = ThreadPoolExecutor(max_workers=n_fly)

    def run_steps(first, count):
        # slot-affine: step i always runs on slot i % n_fly, one step per slot at a time
        chains = [[j for j in range(first, first + count) if j % n_fly == s] for s in range(n_fly)]
        futs = [pool.submit(lambda ch=ch: [step(j) for j in ch]) for ch in chains]
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
    t0 = time.time()
    outs = run_steps(args.warmup, args.steps)
    barrier()
    dt = time.time() - t0
    ka = kb = 0.0
    la = lb = 0
    bif_ms = 0.0
    for out in outs:
        tm = out["result"].timing
        ka += tm["kernel_a_ms"]; kb += tm["kernel_b_ms"]; la += tm["launches_a"]; lb += tm["launches_b"]
        bif_ms += tm["host_bif_ms"]
        assert int(out["result"].stats[:, 0].max()) == 0, "simulator reported error bits"
    out = outs[-1]
    # secondary metric of BASELINE.json: DynUNet-S training images/s at 1x1216x1216, bf16, on the MFMA convolution
    # path; reported so the gap to the 200 imgs/s target is tracked, it is NOT part of `value`.
    train_info = None
    if not args.no_train:
        for g_ in gens:
            g_.close()
        gens = []
        torch.cuda.empty_cache()
        train_info = unet_train_bench(dev, args.train_batch, dist, world)
    # BASELINE.json configs[4]: on-the-fly simulation + rasterisation + GPU augmentation feeding the same training step
    e2e_info = e2e_gan_info = None
    if not args.no_train and not args.no_end_to_end:
        import train_synthetic
        torch.cuda.empty_cache()
        e2e_info = train_synthetic.run(steps=160, batch=args.train_batch, gen_batch=128, seed0=500000, log=False)   # 5 generator batches: past the queue-filling transient
        # configs[4] proper: the same stream feeding the joint GAN contrast-adaptation + segmentation step (G, D at 304^2, S at 1216^2)
        torch.cuda.empty_cache()
        e2e_gan_info = train_synthetic.run(steps=64, batch=args.train_batch, gen_batch=128, seed0=600000, log=False, gan=True)

    dt = sharding.max_over_ranks(dt, dist, dev)

    if rank == 0:
        value = world * B * args.steps / dt
        # dominant kernel: the persistent simulator kernel (one launch per batch runs all iterations of every
        # sample; OCTA_SIM_LOCKSTEP=1 selects the two-launches-per-iteration form, then launch A or B)
        if la == 0:
            dom_ms, dom_n, dom_name = kb, lb, "sim_persistent_kernel"
            bytes_per_launch = ALGO_BYTES_PER_SAMPLE * B
            note = ("one launch = 250 dependent growth iterations of 128 independent samples, one workgroup each; "
                    "dependency/latency-bound (ordered passes, pow chains), not HBM-bound; see DESIGN.md")
        else:
            dom_ms, dom_n, dom_name = (kb, lb, "sim_iter_b_kernel") if kb >= ka else (ka, la, "sim_iter_a_kernel")
            n_iter = max(lb // max(args.steps, 1), 1)
            bytes_per_launch = ALGO_BYTES_PER_SAMPLE / (2.0 * n_iter) * B
            note = "lock-step form: 250 dependent iterations x 2 launches; latency-bound, see DESIGN.md"
        achieved = bytes_per_launch / (dom_ms / max(dom_n, 1) * 1e-3) / 1e9
        line = {
            "metric": "synthetic OCTA samples/sec (graph + 304x304 image + 1216x1216 label triples)",
            "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"configs[1]: {B}-sample vessel-graph batch (docker/vessel_graph_gen_docker_config.yml, "
                                   f"I=100+150, N=2000) + tree2img rasterise 304x304 image and 1216x1216 label",
                       "batch_per_gpu": B, "steps_in_flight": n_fly, "parallelism": f"sample-sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_per_launch(dom_name),
                         "traffic_unit": "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, profiles/r01_bench_pmc_summary.csv)",
                         "algorithmic_bytes_per_launch": bytes_per_launch,
                         "avg_launch_ms": dom_ms / max(dom_n, 1), "launches": dom_n,
                         "note": note},
            "kernel_ms_per_step": {"sim_iter_a": ka / args.steps, dom_name if la == 0 else "sim_iter_b": kb / args.steps,
                                   "host_bifurcation_callback": bif_ms / args.steps},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg)
        line["unet_train"] = train_info
        line["end_to_end_train"] = e2e_info
        line["end_to_end_gan_seg_train"] = e2e_gan_info
        print(json.dumps(line))
    for g_ in gens:
        g_.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
