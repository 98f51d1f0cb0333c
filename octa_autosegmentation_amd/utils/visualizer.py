"""Run directories and files of the training loop (reference utils/visualizer.py:17-328), the parts that are formats rather
than plots: `<save_dir>/<YYYYmmdd_HHMMSS>/` with config.yml, metrics.csv (header `epoch,<titles...>`, one row per epoch),
architecture.txt, checkpoints/<prefix>_model.pth = torch.save({'epoch','model','optimizer','config'}), resume by cloning the
log and the checkpoints into a fresh run directory, `get_max_of_metric`. Sample figures are written as plain side-by-side
PNG strips by the native encoder (matplotlib figures are out of scope)."""
import csv
import datetime
import os
from shutil import copyfile

import numpy as np
import torch
import yaml

from . import checkpoints
from .enums import Phase, Task


def _to_u8(t):
    return (t.squeeze().detach().float().clip(0, 1).cpu().numpy() * 255).astype(np.uint8)


def _save_strip(path, images):
    """images: list of 2-D uint8 arrays of possibly different sizes -> one row, nearest-neighbour scaled to the tallest."""
    from .. import _native
    h = max(a.shape[0] for a in images)
    cols = []
    for a in images:
        if a.shape[0] != h:
            f = h // a.shape[0] if h % a.shape[0] == 0 else None
            a = np.kron(a, np.ones((f, f), np.uint8)) if f else a[(np.arange(h) * a.shape[0] // h)][:, (np.arange(int(a.shape[1] * h / a.shape[0])) * a.shape[0] // h)]
        cols.append(a)
        cols.append(np.full((h, 4), 128, np.uint8))
    strip = np.ascontiguousarray(np.concatenate(cols[:-1], axis=1))
    _native.check(_native.lib().octa_png_write_gray8(path.encode(), strip.ctypes.data, strip.shape[1], strip.shape[0], -1), "octa_png_write_gray8")
    return path


def plot_single_image(save_dir: str, input: torch.Tensor, name: str = None):
    """test.py's prediction file: `<save_dir>/<name without extension>.png` = uint8(input * 255) (visualizer.py:330-339)."""
    from .. import _native
    a = np.ascontiguousarray((input.squeeze().detach().float().cpu().numpy() * 255).astype(np.uint8))
    if a.ndim != 2:
        raise NotImplementedError("3-D predictions (nifti) are outside the MI355X hot path")
    path = os.path.join(save_dir, ".".join(name.split(".")[:-1]) + ".png")
    _native.check(_native.lib().octa_png_write_gray8(path.encode(), a.ctypes.data, a.shape[1], a.shape[0], -1), "octa_png_write_gray8")
    return path


def plot_sample(save_dir: str, input: torch.Tensor, pred: torch.Tensor, truth: torch.Tensor = None, path: str = None, suffix: str = None,
                **unused) -> str:
    suffix = "_" + suffix if suffix else ""
    imgs = [_to_u8(input), _to_u8(pred)] + ([_to_u8(truth)] if truth is not None else [])
    return _save_strip(os.path.join(save_dir, f"sample{suffix}.png"), imgs)


class Visualizer:
    def __init__(self, config: dict, continue_train=False, epoch="latest") -> None:
        self.config = config
        self.save_to_disk: bool = config["Output"]["save_to_disk"]
        self.save_to_tensorboard: bool = config["Output"].get("save_to_tensorboard", False)
        os.makedirs(config["Output"]["save_dir"], exist_ok=True)
        self.track_record = list()
        self.epochs = []
        self.log_file_path = None
        self.start_epoch = int(epoch) if str(epoch).isnumeric() else None
        stamp = lambda: datetime.datetime.now().strftime("%Y%m%d_%H%M%S")
        if continue_train:
            old = config["Output"]["save_dir"]
            name = old.split("/")[-1]
            self.save_dir = os.path.join(old[:-len(name)], stamp())
            os.mkdir(self.save_dir)
            os.mkdir(os.path.join(self.save_dir, "checkpoints"))
            self._copy_log_file(old, self.save_dir, self.start_epoch)
            self._copy_checkpoints(old, self.save_dir, epoch)
            with open(self.log_file_path, newline="") as csvfile:
                for row in csv.DictReader(csvfile):
                    items = list(row.items())
                    if config["General"]["task"] == Task.GAN_VESSEL_SEGMENTATION:
                        d = {"loss": {k: float(v) for k, v in items[1:-2]}, "metric": {k: float(v) for k, v in items[-2:]}}
                    else:
                        d = {"loss": {k: float(v) for k, v in items[1:3]}, "metric": {k: float(v) for k, v in items[3:]}}
                    self.track_record.append(d)
                    self.epochs.append(int(row["epoch"]))
                    if str(epoch).isnumeric() and self.epochs[-1] > int(epoch):
                        break
        else:
            while True:
                self.save_dir = os.path.join(config["Output"]["save_dir"], stamp())
                if not os.path.exists(self.save_dir):
                    os.mkdir(self.save_dir)
                    break
        config["Output"]["save_dir"] = self.save_dir
        config.setdefault(Phase.TEST.value, {})
        config[Phase.TEST]["save_dir"] = os.path.join(self.save_dir, Phase.TEST.value)
        config[Phase.TEST]["model_path"] = os.path.join(self.save_dir, "best_model.pth")
        with open(os.path.join(self.save_dir, "config.yml"), "w") as f:
            yaml.dump(config, f)

    def _copy_log_file(self, old_dir, new_dir, epoch=None):
        old_log = os.path.join(old_dir, "metrics.csv")
        self.log_file_path = os.path.join(new_dir, "metrics.csv")
        if epoch:
            with open(old_log) as f:
                rows = f.readlines()[0:epoch + 1]
            with open(self.log_file_path, "w") as f:
                f.writelines(rows)
        else:
            copyfile(old_log, self.log_file_path)

    def _copy_checkpoints(self, old_dir, new_dir, epoch="latest"):
        """Every checkpoint file of tag `epoch` (and `best`) moves to the new run directory. (The reference copies
        `<epoch>_{G,D,S}_model.pth` / `best_model.pth`, names its own train.py does not write; what train.py writes is
        `<tag>_<net>_model.pth` + `<tag>_<optimizer>_model.pth`, base_model_abc.py:67-85 reads them from save_dir.)"""
        src = os.path.join(old_dir, "checkpoints")
        for name in sorted(os.listdir(src)) if os.path.isdir(src) else []:
            if name.startswith(f"{epoch}_") or name.startswith("best_"):
                copyfile(os.path.join(src, name), os.path.join(new_dir, "checkpoints", name))

    def plot_losses_and_metrics(self, metric_groups: dict, epoch: int):
        self.track_record.append({title: metrics for title, metrics in metric_groups.items()})
        if self.log_file_path is None:
            self.log_file_path = os.path.join(self.save_dir, "metrics.csv")
            with open(self.log_file_path, "w+") as file:
                csv.writer(file).writerow(["epoch", *[t for v in metric_groups.values() for t in v]])
        self.epochs.append(epoch)
        with open(self.log_file_path, "a", newline="") as file:
            csv.writer(file).writerow([epoch, *[x for v in metric_groups.values() for x in v.values()]])

    def save_model_architecture(self, model: torch.nn.Module, input: torch.Tensor):
        rows = [(n, p.numel()) for n, p in model.named_parameters() if p.requires_grad]
        with open(os.path.join(self.save_dir, "architecture.txt"), "w+") as f:
            f.write(str(model) + "\n")
            for n, c in rows:
                f.write(f"{n}\t{c}\n")
            f.write(f"Total Trainable Params: {sum(c for _, c in rows)}")

    def save_model(self, model, optimizer, epoch: int, config: dict, prefix: str = "") -> str:
        return checkpoints.save_model(self.save_dir, model, optimizer, epoch, config, prefix)

    def log_model_params(self, model, epoch: int):
        pass                                            # TensorBoard histograms: `self.tb` is never constructed in the reference

    def get_max_of_metric(self, metric_type: str, metric_name: str):
        vals = [m[metric_type][metric_name] for m in self.track_record]
        return max(vals), int(np.argmax(vals))

    def plot_sample(self, input, pred, truth=None, path=None, suffix: str = None, **unused) -> str:
        return plot_sample(self.save_dir, input, pred, truth, path, suffix)

    def plot_gan_seg_sample(self, real_A, fake_B, fake_B_seg, real_B, idt_B, real_B_seg, path_A=None, path_B=None, suffix="", **unused):
        imgs = [_to_u8(t) for t in (real_A, fake_B, fake_B_seg, real_B, idt_B, real_B_seg) if t is not None]
        return _save_strip(os.path.join(self.save_dir, f"sample_{suffix}.png"), imgs)
