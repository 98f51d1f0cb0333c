"""Drop-in for the reference's train.py on MI355X (reference train.py:29-230): same flags (`--config_file --start_epoch --epoch
--split --save_latest --num_workers`, dotted `--A.B.c value` overrides), same config schema, same run directory
(`<Output.save_dir>/<timestamp>/{config.yml, metrics.csv, architecture.txt, checkpoints/<tag>_<net|optimizer>_model.pth}`),
same epoch structure (train, LR schedule step, validation every `val_interval`, `latest` / `<epoch>` / `best` checkpoints).

What differs is where the work happens: samples are loaded, rasterised and augmented on the GPU (data/), the model behind
ModelInterface runs the hand-written HIP kernels (models/), losses stay device tensors until the epoch ends (one host
read per epoch instead of one `.item()` per loss and step). Multi-GPU: launch with
`python -m torch.distributed.run --nproc-per-node N train.py ...` -- one process per GPU, every rank takes its share of each
epoch's batches, gradients are averaged with one RCCL all-reduce per optimiser and step, rank 0 writes the files."""
import argparse
import datetime
import os
import sys
import time
from copy import deepcopy
from random import randint
from shutil import copyfile

import torch
import yaml

ROOT = os.path.dirname(os.path.abspath(__file__))
LAST_RUN = {}        # filled by train(): per-epoch images per second of the last run in this process (read by bench.py)


def set_determinism(seed):
    """monai.utils.set_determinism: Python, numpy and torch generators."""
    import random
    import numpy as np
    random.seed(seed)
    np.random.seed(seed % (2 ** 32))
    torch.manual_seed(seed)


def _distributed():
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        dist.init_process_group(backend="nccl" if torch.cuda.is_available() else "gloo",
                                device_id=torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None)
    return (dist.get_rank(), dist.get_world_size(), dist) if world > 1 else (0, 1, None)


def train(args: argparse.Namespace, config: dict):
    from octa_autosegmentation_amd.data.image_dataset import get_dataset, get_post_transformation
    from octa_autosegmentation_amd.models.model import define_model
    from octa_autosegmentation_amd.models.networks import init_weights
    from octa_autosegmentation_amd.utils.aside import join_aside
    from octa_autosegmentation_amd.utils.enums import Phase
    from octa_autosegmentation_amd.utils.metrics import MetricsManager
    from octa_autosegmentation_amd.utils.visualizer import Visualizer
    rank, world, dist = _distributed()
    sys.setswitchinterval(0.0005)      # the loader thread and this one share the GIL: hand it over in 0.5 ms slices, not 5 ms ones
    for phase in Phase:
        if phase not in config:
            continue
        for k in (config[phase].get("data") or {}).keys():
            if not config[phase]["data"][k].get("split", ".txt").endswith(".txt"):
                assert bool(args.split), "You have to specify a split!"
                config[phase]["data"][k]["split"] = config[phase]["data"][k]["split"] + args.split + ".txt"

    max_epochs = config[Phase.TRAIN]["epochs"]
    val_interval = config[Phase.TRAIN].get("val_interval") or 1
    save_interval = config[Phase.TRAIN].get("save_interval") or 100
    scaler = torch.amp.GradScaler("cuda", enabled=False)        # bf16 autocast needs no loss scaling; kept for the interface
    if world > 1:
        config["General"]["device"] = f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cpu"
    device = torch.device(config["General"].get("device") or "cpu")
    if device.type == "cuda":
        torch.cuda.set_device(device)
    if rank == 0:
        visualizer = Visualizer(config, args.start_epoch > 0, epoch=args.epoch)
    if world > 1:                                               # every rank resolves checkpoints against rank 0's run directory
        box = [config["Output"]["save_dir"]]
        dist.broadcast_object_list(box, src=0)
        config["Output"]["save_dir"] = box[0]

    train_loader = get_dataset(config, Phase.TRAIN, num_workers=args.num_workers)
    train_loader.shard = (rank, world)
    if world > 1:
        # only the batch permutation is shared between ranks (its own generator, same seed everywhere); python's, numpy's and torch's
        # global streams and every random transform continue from seed + rank, so ranks draw different flips, rotations, real_B /
        # background picks and noise for their k-th batch
        seed = int(config["General"].get("seed", 42))
        train_loader.perm_generator = torch.Generator().manual_seed(seed)
        set_determinism(seed + rank)
        train_loader.reseed_augmentations(seed + rank)
    post_transformations_train = get_post_transformation(config, Phase.TRAIN)
    if Phase.VALIDATION in config:
        val_loader = get_dataset(config, Phase.VALIDATION, num_workers=args.num_workers)
        post_transformations_val = get_post_transformation(config, Phase.VALIDATION)
    else:
        val_loader = None
        print("No validation config. Skipping validation steps.", file=sys.stderr)

    model = define_model(deepcopy(config), phase=Phase.TRAIN)
    model.initialize_model_and_optimizer(None, init_weights, config, args, scaler, phase=Phase.TRAIN)
    if rank == 0:
        visualizer.save_model_architecture(model, None)
    metrics = MetricsManager(phase=Phase.TRAIN)
    if args.start_epoch > 0 and rank == 0:
        best_metric, best_metric_epoch = visualizer.get_max_of_metric("metric", metrics.get_comp_metric(Phase.VALIDATION))
    else:
        best_metric, best_metric_epoch = -1, -1

    total_start = time.time()
    LAST_RUN.clear()
    LAST_RUN["imgs_per_s"] = []
    for epoch in range(args.start_epoch, max_epochs):
        t_epoch = time.time()
        epoch_metrics = {"loss": dict()}
        model.train()
        step = 0
        save_best = False
        running = None                                           # device-side running sums of the losses: one read per epoch
        for mini_batch in train_loader:
            step += 1
            outputs, losses = model.perform_training_step(mini_batch, scaler, post_transformations_train, device)
            with torch.autocast(device_type=device.type, dtype=torch.bfloat16, enabled=device.type == "cuda"):
                model.compute_metric(outputs, metrics)
            vals = torch.stack([torch.as_tensor(v, device=device).detach().float().reshape(()) for v in losses.values()])
            running = vals if running is None else running + vals
        n_img = step * train_loader.batch_size * world
        for lr_scheduler in model.lr_schedulers:
            lr_scheduler.step()
        join_aside(device)                                       # the metric kernels of the epoch (side stream) before their scores are read
        sums = running.cpu().tolist()
        epoch_metrics["loss"] = {f"train_{k}": v / step for k, v in zip(losses.keys(), sums)}
        epoch_metrics["metric"] = metrics.aggregate_and_reset(prefix=Phase.TRAIN)
        main_loss = list(losses.keys())[0]
        LAST_RUN["imgs_per_s"].append(n_img / (time.time() - t_epoch))
        print(f"epoch {epoch + 1}/{max_epochs}: train {main_loss} {epoch_metrics['loss']['train_' + main_loss]:.4f}, "
              f"{LAST_RUN['imgs_per_s'][-1]:.1f} imgs/s", file=sys.stderr, flush=True)
        train_sample_path = val_sample_path = None
        if rank == 0 and (args.save_latest or (epoch + 1) % save_interval == 0):
            train_sample_path = model.plot_sample(visualizer, mini_batch, outputs, suffix="train_latest")

        if val_loader is not None and (epoch + 1) % val_interval == 0:
            model.eval()
            vsum, vstep, vkeys = None, 0, []
            with torch.no_grad():
                for val_mini_batch in val_loader:
                    vstep += 1
                    with model.autocast():
                        outputs, losses = model.inference(val_mini_batch, post_transformations_val, device=device, phase=Phase.VALIDATION)
                        model.compute_metric(outputs, metrics)
                    vkeys = list(losses.keys())
                    vals = torch.stack([v.detach().float().reshape(()) for v in losses.values()])
                    vsum = vals if vsum is None else vsum + vals
            for k, v in zip(vkeys, (vsum / vstep).cpu().tolist()):
                epoch_metrics["loss"][f"val_{k}"] = v
            epoch_metrics["metric"].update(metrics.aggregate_and_reset(prefix=Phase.VALIDATION))
            metric_comp = epoch_metrics["metric"][metrics.get_comp_metric(Phase.VALIDATION)]
            if metric_comp > best_metric:
                best_metric, best_metric_epoch, save_best = metric_comp, epoch, True
            if rank == 0 and (args.save_latest or save_best or (epoch + 1) % save_interval == 0):
                val_sample_path = model.plot_sample(visualizer, val_mini_batch, outputs, suffix="val_latest")

        if rank == 0:
            for p in (train_sample_path, val_sample_path):
                if p is None:
                    continue
                if (epoch + 1) % save_interval == 0:
                    copyfile(p, p.replace("latest", str(epoch + 1)))
                if save_best:
                    copyfile(p, p.replace("latest", "best"))
            if args.save_latest or save_best or (epoch + 1) % save_interval == 0:
                def keep(path):
                    if (epoch + 1) % save_interval == 0:
                        copyfile(path, path.replace("latest", str(epoch + 1)))
                    if save_best:
                        copyfile(path, path.replace("latest", "best"))
                for optimizer_name in model.optimizer_mapping.keys():
                    keep(visualizer.save_model(None, getattr(model, optimizer_name), epoch + 1, config, f"latest_{optimizer_name}"))
                for model_names in model.optimizer_mapping.values():
                    for model_name in model_names:
                        keep(visualizer.save_model(getattr(model, model_name), None, epoch + 1, config, f"latest_{model_name}"))
            visualizer.plot_losses_and_metrics(epoch_metrics, epoch)
            visualizer.log_model_params(model, epoch)
        if dist is not None:
            dist.barrier()

    train_loader.close()
    if val_loader is not None:
        val_loader.close()
    print(f"Finished training after {str(datetime.timedelta(seconds=time.time() - total_start))}.", file=sys.stderr)
    if best_metric_epoch > -1:
        print(f"Best metric: {best_metric} at epoch: {best_metric_epoch}.", file=sys.stderr)
    return config["Output"]["save_dir"]


def main(argv=None):
    parser = argparse.ArgumentParser(description="")
    parser.add_argument("--config_file", type=str, required=True)
    parser.add_argument("--start_epoch", type=int, default=0)
    parser.add_argument("--epoch", type=str, default="latest")
    parser.add_argument("--split", type=str, default="")
    parser.add_argument("--save_latest", type=bool, default=True, help="If true, save a checkpoint and visuals after each epoch under the tag 'latest'.")
    parser.add_argument("--num_workers", type=int, default=None, help="0: samples are prepared inline; otherwise one loader thread prefetches on its own HIP stream.")
    args, unknown = parser.parse_known_args(argv)
    path = os.path.abspath(args.config_file)
    assert os.path.isfile(path), f"Your provided config path {args.config_file} does not exist!"
    with open(path, "r") as stream:
        config = yaml.safe_load(stream)
    from octa_autosegmentation_amd.utils.config_overrides import apply_cli_overrides_from_unknown_args
    apply_cli_overrides_from_unknown_args(config, unknown)
    if "seed" not in config["General"]:
        config["General"]["seed"] = randint(0, int(1e6))
        if int(os.environ.get("WORLD_SIZE", "1")) > 1:          # every rank shuffles with rank 0's seed
            _, _, dist = _distributed()
            box = [config["General"]["seed"]]
            dist.broadcast_object_list(box, src=0)
            config["General"]["seed"] = box[0]
    set_determinism(seed=config["General"]["seed"])
    return train(args, config)


if __name__ == "__main__":
    main()
