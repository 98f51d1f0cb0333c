"""On-the-fly training: HIP vessel simulation -> rasterisation -> GPU augmentation -> U-Net step, all on one MI355X.

This is BASELINE.json configs[4] for the segmentation network ("on-the-fly HIP vessel simulation feeding training, overlapped
gen/compute"): instead of the reference's loader (CSV graphs on disk, `LoadGraphAndFilterByRandomRadiusd` + MONAI transforms in
CPU workers, configs/config_ves_seg-S.yml:28-102, train.py:29-203) a generator thread keeps simulating seeded samples in
small batches on its own HIP stream and hands uint8 image / label batches over in HBM; the training thread applies the
config's augmentation list on the GPU (data/gpu_augment.py) and runs the training step (models/segmentation_trainer.py).
Multi-GPU: launch with torch.distributed.run -- every rank generates and trains on its own seeds, gradients are
all-reduced with RCCL (the trainer's flat all-reduce).

  python train_synthetic.py --steps 50 --batch 4 --gen-batch 256
"""
import argparse
import json
import os
import queue
import sys
import threading
import time

import numpy as np

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEG_CONFIG = {   # configs/config_ves_seg-S.yml, the parts this script uses
    "General": {"amp": True, "model": {"name": "DynUNet", "spatial_dims": 2, "in_channels": 1, "out_channels": 1,
                                       "kernel_size": [3, 3, 3, 3, 3], "strides": [1, 2, 2, 2, 1], "upsample_kernel_size": [1, 2, 2, 2, 1]}},
    "Train": {"lr": 1e-4, "loss": "DiceBCELoss", "epochs": 30, "epochs_decay": 10, "batch_size": 4,
              "data_augmentation": [
                  {"name": "LoadGraphAndFilterByRandomRadiusd", "keys": ["image", "label"], "image_resolutions": [[304, 304], [1216, 1216]],
                   "min_radius": [0, 0.0033], "max_dropout_prob": 0},
                  {"name": "ScaleIntensityd", "keys": ["image", "label"], "minv": 0, "maxv": 1},
                  {"name": "EnsureChannelFirstd", "keys": ["image", "label"], "strict_check": False, "channel_dim": "no_channel"},
                  {"name": "Resized", "keys": ["image", "label"], "spatial_size": [1216, 1216], "mode": "bilinear"},
                  {"name": "RandFlipd", "keys": ["image", "label"], "prob": 0.5, "spatial_axis": [0, 1]},
                  {"name": "RandRotate90d", "keys": ["image", "label"], "prob": 0.75},
                  {"name": "RandRotated", "keys": ["image", "label"], "prob": 1, "range_x": 0.17453292519943295, "padding_mode": "zeros"},
                  {"name": "AsDiscreted", "keys": ["label"], "threshold": 0.1},
                  {"name": "CastToTyped", "keys": ["image", "label"], "dtype": "dtype"}]},
}


GAN_CONFIG = {   # configs/config_gan_ves_seg.yml, the parts this script uses
    "General": {"amp": True, "model": {"name": "GanSegModel", "model_g": {"name": "resnetGenerator9"}, "model_d": {"name": "patchGAN70x70"},
                                       "model_s": SEG_CONFIG["General"]["model"], "upshape": (1216, 1216),
                                       "compute_identity": False, "compute_identity_seg": True}},
    "Train": {"lr": 2e-4, "loss_dg": "LSGANLoss", "loss_s": "DiceBCELoss", "batch_size": 4,
              "data_augmentation": [   # the entries applied to real_A / real_A_seg after the graph loader (same draws for both keys)
                  {"name": "ScaleIntensityd", "keys": ["real_A", "real_A_seg"], "minv": 0, "maxv": 1},
                  {"name": "RandFlipd", "keys": ["real_A", "real_A_seg"], "prob": 0.5, "spatial_axis": [0, 1]},
                  {"name": "RandRotate90d", "keys": ["real_A", "real_A_seg"], "prob": 0.75},
                  {"name": "RandRotated", "keys": ["real_A", "real_A_seg"], "prob": 1, "range_x": 0.17453292519943295, "padding_mode": "zeros"},
                  {"name": "AsDiscreted", "keys": ["real_A_seg"], "threshold": 0.1}]},
}


def load_sim_config():
    from octa_autosegmentation_amd.utils import configs
    return configs.load_generator_config()


def run(steps, batch, gen_batch, seed0=0, warmup=3, log=True, gan=False):
    import torch
    import torch.distributed as dist
    from octa_autosegmentation_amd import pipeline
    from octa_autosegmentation_amd.data.gpu_augment import GpuSegAugmentation
    from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
    from octa_autosegmentation_amd.utils import sharding
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dev = torch.device("cuda", torch.cuda.current_device())
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(backend="nccl", device_id=dev)
    if gan:
        # BASELINE configs[4] proper: the on-the-fly stream feeds the joint GAN contrast-adaptation + segmentation step
        # (configs/config_gan_ves_seg.yml; graph loader with min_radius [0, 0]). real_B (real OCTA scans) and the background
        # tiles are not in the repository: uniform-noise stand-ins of the same shape (data = synthetic, as everywhere here).
        from octa_autosegmentation_amd.models.gan_seg_trainer import GanSegTrainer
        aug_cfg = GAN_CONFIG["Train"]["data_augmentation"]
        gen = pipeline.TripleGenerator(load_sim_config(), gen_batch, label_resolution=[1216, 1216], label_min_radius=0, image_mode="loader")
        trainer = GanSegTrainer(GAN_CONFIG, dev)
    else:
        aug_cfg = SEG_CONFIG["Train"]["data_augmentation"]
        loader = aug_cfg[0]
        gen = pipeline.TripleGenerator(load_sim_config(), gen_batch, label_resolution=loader["image_resolutions"][1],
                                       label_min_radius=loader["min_radius"][1], image_mode="loader", image_min_radius=loader["min_radius"][0])
        trainer = SegmentationTrainer(SEG_CONFIG, dev)
    aug = GpuSegAugmentation(aug_cfg, seed=1234 + rank)
    noise = torch.Generator(device=dev).manual_seed(99 + rank)
    q = queue.Queue(maxsize=2)
    stop = threading.Event()
    gen_stream = torch.cuda.Stream()
    failure = []                                                            # the producer's exception, re-raised by the consumer

    def produce():
        try:
            torch.cuda.set_device(dev)
            i = 0
            with torch.cuda.stream(gen_stream):
                out = None
                while not stop.is_set():
                    if i > 0:
                        turns.ask(stop)                                     # the trainer steps aside at its next step boundary (see `turns` below)
                    try:
                        out = gen.generate(sharding.rank_seeds(rank, i, gen_batch, base=seed0))
                    finally:
                        turns.hand_back()
                    ready = torch.cuda.Event()
                    ready.record(gen_stream)
                    item = (out["image"], out["label_grey"], ready)
                    while not stop.is_set():
                        try:
                            q.put(item, timeout=0.1)
                            break
                        except queue.Full:
                            pass
                    i += 1
        except BaseException as e:                                          # noqa: BLE001 -- handed to the consumer, which re-raises
            failure.append(e)

    # Taking turns (round 5, utils/turns.py): the persistent kernel's workgroups hold every CU's LDS and registers, so a training step that shares
    # the GPU with a generator launch takes 112 instead of 17 ms while the launch itself stretches from 410 to 729 ms
    # (tools/exp_r05_e2e_idle.sh). The producer asks for the GPU before a launch; the trainer answers at its next step boundary with its queued
    # kernels drained and waits until the simulator call has returned (the rasterisation that follows shares the GPU with training as
    # before). OCTA_E2E_BURST=0: both share the GPU (rounds 1-4).
    from octa_autosegmentation_amd.utils.turns import GpuTurns
    turns = GpuTurns(enabled=os.environ.get("OCTA_E2E_BURST", "1") != "0")

    # Several ranks (round 6, advisor): a rank that steps aside stalls EVERY rank at the step's gradient all-reduce, so turns given away at each
    # rank's own moment would add up (world x 410 ms per generator batch). With world > 1 a turn is only given at the step where the rank needs its
    # next generator batch -- the same step on every rank, since all ranks consume gen_batch / batch steps per batch -- and while waiting for that
    # batch: the ranks' stalls then coincide instead of queueing behind each other. OCTA_E2E_ALIGN=0/1 overrides.
    turns.aligned = os.environ.get("OCTA_E2E_ALIGN", "1" if world > 1 else "0") == "1"

    def step_aside_if_asked(at_batch_boundary=True):
        turns.step_aside_if_asked(lambda: torch.cuda.current_stream().synchronize(), lambda: th.is_alive() and not failure, at_boundary=at_batch_boundary)

    def next_batch():
        """Blocks for the producer's next batch; a dead producer is an error here, not a silent stall (the reference swallows
        worker exceptions, generate_vessel_graph.py:127-129; SURVEY section 5 asks the build to surface them)."""
        while True:
            if failure:
                raise RuntimeError("the generator thread failed") from failure[0]
            step_aside_if_asked()              # the producer may be waiting for the GPU to make the very batch this call waits for
            try:
                return q.get(timeout=0.02)
            except queue.Empty:
                if not th.is_alive() and not failure:
                    raise RuntimeError("the generator thread ended without handing over a batch")

    th = threading.Thread(target=produce, name="octa-generator")
    th.start()
    images = labels = None
    pos = 0
    losses = []
    t0 = None
    try:
        for step in range(warmup + steps):
            if step == warmup:
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t0 = time.time()
            step_aside_if_asked(at_batch_boundary=images is None or pos + batch > images.shape[0])
            if images is None or pos + batch > images.shape[0]:
                images, labels, ready = next_batch()
                # the batch was written on the generator's stream and its blocks belong to that stream's pool: order the
                # training stream after the writes, and keep the allocator from recycling the blocks under queued readers
                cur = torch.cuda.current_stream()
                cur.wait_event(ready)
                images.record_stream(cur)
                labels.record_stream(cur)
                pos = 0
            mb = aug(images[pos:pos + batch].contiguous(), labels[pos:pos + batch].contiguous())
            pos += batch
            if gan:
                a = mb["image"]
                bg = torch.rand(a.shape, device=dev, generator=noise) * torch.rand(a.shape, device=dev, generator=noise)
                real_a = torch.maximum(a, bg)                      # AddRandomBackgroundNoised (data_transforms.py:506-516)
                real_b = torch.rand(a.shape, device=dev, generator=noise)
                _, l = trainer.perform_training_step({"real_A": real_a, "real_B": real_b, "real_A_seg": mb["label"]})
                losses.append(l["S"].detach())
            else:
                _, l = trainer.perform_training_step({"image": mb["image"], "label": mb["label"]})
                losses.append(l[trainer.loss_name].detach() if hasattr(l[trainer.loss_name], "detach") else l[trainer.loss_name])
        torch.cuda.synchronize()
        dt = sharding.max_over_ranks(time.time() - t0, dist if world > 1 else None, dev)
    finally:
        stop.set()
        th.join()
        gen.close()
    if failure:                                                             # e.g. the batch being generated when the loop ended
        raise RuntimeError("the generator thread failed") from failure[0]
    res = {"metric": ("end-to-end on-the-fly GAN-seg training imgs/s (simulate + rasterise + augment + G/D @304^2 + DynUNet-S @1216^2 step)" if gan else
                      "end-to-end on-the-fly training imgs/s (simulate + rasterise + augment + DynUNet-S step @1216^2)"),
           "value": world * batch * steps / dt, "unit": "imgs/s", "n_gpus": world, "steps": steps, "ms_per_step": dt / steps * 1e3,
           "batch_per_gpu": batch, "generator_batch": gen_batch, "first_loss": float(losses[0]), "last_loss": float(losses[-1])}
    if log and rank == 0:
        print(json.dumps(res))
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--gen-batch", type=int, default=512, help="samples per generator launch: 512 fills every workgroup slot of an MI355X (two per CU), where the persistent kernel is cheapest per sample")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--gan", action="store_true", help="feed the GAN-seg trainer (configs[4]) instead of the segmentation trainer")
    a = ap.parse_args()
    run(a.steps, a.batch, a.gen_batch, warmup=a.warmup, gan=a.gan)
