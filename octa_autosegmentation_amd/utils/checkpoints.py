"""On-disk formats of the training loop, so that a run can move between the reference trainer and this one mid-way
(SURVEY.md 8f rank 3). Restated from the reference's behaviour, not its code:

* checkpoints (utils/visualizer.py:225-238, train.py:175-190): `<save_dir>/checkpoints/<prefix>_model.pth`, a torch.save'd
  dict {'epoch', 'model', 'optimizer', 'config'}; per epoch one file per OPTIMIZER (`latest_<optimizer_name>_model.pth`,
  'model' None) and one per NETWORK (`latest_<net_name>_model.pth`, 'optimizer' None); copies named `<epoch>_...` every
  save_interval epochs and `best_...` when the validation metric improved.
* resume (models/base_model_abc.py:67-85): given `.../<prefix>_model.pth`, network weights come from
  `<prefix>_<net_name>_model.pth`, optimizer states from the same dict or from `<prefix>_<optimizer_name>.pth`'s
  sibling `<prefix>_<optimizer_name>_model.pth` (what train.py actually writes).
* metrics.csv (utils/visualizer.py:74-79, 135-137): header `epoch,<title>...`, one row per epoch, titles flattened over the
  metric groups in insertion order.
"""
import csv
import os
import shutil

import torch
import yaml


def save_model(save_dir, model, optimizer, epoch, config, prefix=""):
    os.makedirs(os.path.join(save_dir, "checkpoints"), exist_ok=True)
    path = os.path.join(save_dir, "checkpoints", f"{prefix}_model.pth")
    sd = None
    if model is not None:
        sd = model.state_dict()
    torch.save({"epoch": epoch, "model": sd, "optimizer": optimizer.state_dict() if optimizer is not None else None, "config": config}, path)
    return path


def save_epoch(save_dir, trainer, epoch, config, save_interval=10, save_best=False):
    """What train.py does at the end of epoch `epoch` (0-based): files are labelled epoch + 1."""
    written = []
    for optimizer_name in trainer.optimizer_mapping:
        p = save_model(save_dir, None, getattr(trainer, optimizer_name), epoch + 1, config, f"latest_{optimizer_name}")
        written.append(p)
        if (epoch + 1) % save_interval == 0:
            shutil.copyfile(p, p.replace("latest", str(epoch + 1)))
        if save_best:
            shutil.copyfile(p, p.replace("latest", "best"))
    for names in trainer.optimizer_mapping.values():
        for net_name in names:
            p = save_model(save_dir, getattr(trainer, net_name), None, epoch + 1, config, f"latest_{net_name}")
            written.append(p)
            if (epoch + 1) % save_interval == 0:
                shutil.copyfile(p, p.replace("latest", str(epoch + 1)))
            if save_best:
                shutil.copyfile(p, p.replace("latest", "best"))
    return written


def load_checkpoint(trainer, model_path, device="cpu"):
    """model_path: `<dir>/checkpoints/<prefix>_model.pth` as passed to the reference's --model_path. Returns the epoch."""
    epoch = None
    for optimizer_name, net_names in trainer.optimizer_mapping.items():
        checkpoint = None
        for net_name in net_names:
            checkpoint = torch.load(model_path.replace("model.pth", f"{net_name}_model.pth"), map_location=device, weights_only=False)
            getattr(trainer, net_name).load_state_dict(checkpoint["model"])
        optimizer = getattr(trainer, optimizer_name)
        if checkpoint is not None and checkpoint.get("optimizer") is not None:
            optimizer.load_state_dict(checkpoint["optimizer"])
        else:
            for cand in (model_path.replace("model.pth", f"{optimizer_name}.pth"), model_path.replace("model.pth", f"{optimizer_name}_model.pth")):
                if os.path.exists(cand):
                    optimizer.load_state_dict(torch.load(cand, map_location=device, weights_only=False)["optimizer"])
                    break
            else:
                raise FileNotFoundError(f"no optimizer state for {optimizer_name} next to {model_path}")
        if checkpoint is not None:
            epoch = checkpoint["epoch"]
    return epoch


class MetricsLog:
    """metrics.csv + config.yml of a run directory."""

    def __init__(self, save_dir, config=None):
        self.save_dir = save_dir
        os.makedirs(save_dir, exist_ok=True)
        self.path = os.path.join(save_dir, "metrics.csv")
        self._started = False
        if config is not None:
            with open(os.path.join(save_dir, "config.yml"), "w") as f:
                yaml.dump(config, f)

    def append(self, epoch, metric_groups):
        """metric_groups: {group title: {metric title: value}} in insertion order."""
        if not self._started:
            with open(self.path, "w+") as f:
                csv.writer(f).writerow(["epoch", *[t for v in metric_groups.values() for t in v]])
            self._started = True
        with open(self.path, "a", newline="") as f:
            csv.writer(f).writerow([epoch, *[x for v in metric_groups.values() for x in v.values()]])
