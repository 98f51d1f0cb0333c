"""Shared behaviour of every trainable model behind ModelInterface (reference models/base_model_abc.py:13-171), for one
process per GPU:

* optimisers: one Adam per entry of `optimizer_mapping` (lr from the config, betas (0.5, 0.999) unless overridden,
  `Train.weight_decay`), LambdaLR constant then linear to zero over `Train.epochs_decay` -- attached the way the reference
  attaches them (every scheduler to the LAST optimiser, base_model_abc.py:63-64 uses the stale loop variable; said once at
  construction when it matters; `Train.lr_scheduler_per_optimizer: true` gives one per optimiser);
* initialisation: He-normal through `init_weights` ('relu' gain for resnet generators, 'leaky_relu' otherwise) or, with
  `args.start_epoch > 0`, resume from `<save_dir>/checkpoints/<epoch>_<net>_model.pth` + `<epoch>_<optimizer>_model.pth`
  (what train.py writes; the reference reads `<epoch>_<optimizer>.pth`, which its own train.py never creates -- both names
  are accepted here);
* inference phases: only the network named by `General.inference` is loaded (`model`, `segmentor`/`S`, `generator`/`G`,
  legacy `<epoch>_model.pth`);
* training step: bf16 autocast (no loss scaling: the GradScaler argument is accepted and left alone, bf16 has fp32's
  exponent range), backward, ONE all-reduce per optimiser over a flat fp32 gradient arena, optimiser step.

Gradient arena (world size > 1, or OCTA_GRAD_ARENA=1): every parameter's `.grad` is a view into one contiguous fp32 buffer per
optimiser. autograd accumulates into the views in place, "zero_grad" is one fill, the data-parallel exchange is one
RCCL all-reduce on the buffer itself -- no per-parameter copies in or out (round 1 copied ~70 tensors each way per step).
"""
import itertools
import os
import sys
from abc import ABC, abstractmethod
from typing import Any, Callable, Dict, Tuple

import torch
import torch.distributed as dist
from torch import nn

from ..utils.aside import aside, join_aside  # noqa: F401  (re-exported: models use `aside`, the entry points `join_aside`)
from ..utils.enums import Phase
from .model_interface_abc import ModelInterface, Output


def _log(msg):
    """Progress notes go to stderr: stdout belongs to the entry points' results (bench.py prints exactly one JSON line)."""
    print(msg, file=sys.stderr)


def _dist_on():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class GradArena:
    """One flat fp32 buffer holding the gradients of a parameter list as views."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.attach()

    def attach(self):
        off = 0
        for p in self.params:
            assert p.dtype == torch.float32, "master parameters are fp32"
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()
        for p in self.params:                         # someone (e.g. zero_grad(set_to_none=True)) dropped a view: re-attach all
            if p.grad is None or p.grad.untyped_storage().data_ptr() != self.flat.untyped_storage().data_ptr():
                self.attach()
                break

    def check_views(self):
        """Every parameter's gradient must still BE its arena view when the exchange starts: a `.grad` autograd replaced (instead
        of accumulating in place) would leave zeros in the buffer and the all-reduce would silently send them."""
        base = self.flat.untyped_storage().data_ptr()
        for p in self.params:
            if p.grad is None or p.grad.untyped_storage().data_ptr() != base:
                raise RuntimeError("GradArena: a parameter's .grad no longer aliases the flat gradient buffer "
                                   f"(shape {tuple(p.shape)}); its gradient would be lost in the all-reduce")

    def all_reduce_mean(self):
        self.check_views()
        if dist.is_available() and dist.is_initialized():      # also at world size 1 (OCTA_GRAD_ARENA=1): same code path, RCCL on the buffer
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            if dist.get_world_size() > 1:
                self.flat.div_(dist.get_world_size())


def load_checkpoint_file(path, device):
    """torch.load of a reference-format checkpoint dict {'epoch','model','optimizer','config'} (utils/visualizer.py:225-238);
    the dict carries the whole config (plain python containers), hence weights_only=False."""
    return torch.load(path, map_location=device, weights_only=False)


class BaseModelABC(nn.Module, ModelInterface, ABC):
    def __init__(self, optimizer_mapping=None, optimizer_configs: Dict[str, dict] = None, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.optimizer_mapping: Dict[str, list] = optimizer_mapping or {"optimizer": []}
        self.optimizer_configs = optimizer_configs or dict()
        self.lr_schedulers = []
        self._arenas: Dict[str, GradArena] = {}
        self.amp = True

    # ---- construction -------------------------------------------------------------------------------------------
    def _nets_of(self, net_names):
        return [self] if len(net_names) == 0 else [getattr(self, n) for n in net_names]

    def initialize_model_and_optimizer(self, init_mini_batch: dict, init_weights: Callable, config: dict, args, scaler,
                                       phase=Phase.TRAIN):
        nets = [getattr(self, n, None) for names in self.optimizer_mapping.values() for n in names]
        if not any(isinstance(n, nn.Module) for n in nets):
            _log(f"Skipping initialization for {list(self.optimizer_mapping.values())}")
            return
        device = torch.device(config["General"].get("device") or "cpu")
        self.amp = bool(config["General"].get("amp")) and device.type == "cuda"
        epoch_tag = getattr(args, "epoch", "latest")
        model_path = os.path.join(config["Output"]["save_dir"], "checkpoints", f"{epoch_tag}_model.pth")
        if phase == Phase.TRAIN:
            tr = config[Phase.TRAIN]
            for optim_name, net_names in self.optimizer_mapping.items():
                params = itertools.chain(*[n.parameters() for n in self._nets_of(net_names)])
                setattr(self, optim_name, torch.optim.Adam(params, **{"lr": tr["lr"], "betas": (0.5, 0.999),
                                                                      "weight_decay": tr.get("weight_decay", 0),
                                                                      **self.optimizer_configs.get(optim_name, {})}))
            max_epochs, decay = tr["epochs"], tr.get("epochs_decay", 0)

            def schedule(step: int):
                return 1 if step < (max_epochs - decay) else (max_epochs - step) * (1 / max(1, decay))

            # The reference builds one LambdaLR per optimiser NAME but hands every one of them the LAST optimiser (base_model_abc.py:63-64:
            # `getattr(self, optim_name)` with the previous loop's stale variable): in a multi-optimiser model (GanSegModel: G, D, S) only
            # the last optimiser (S) ever decays, G and D keep their initial learning rate. Identical results need the same behaviour, so
            # it is the DEFAULT; `Train.lr_scheduler_per_optimizer: true` attaches one scheduler to each optimiser instead. With one
            # optimiser (the segmentation configs) or `epochs_decay: 0` (configs/config_gan_ves_seg.yml) the two are the same thing.
            names = list(self.optimizer_mapping)
            per_optimizer = bool(tr.get("lr_scheduler_per_optimizer", False))
            if len(names) > 1 and decay > 0:
                _log(f"NOTE: {len(names)} optimisers with Train.epochs_decay = {decay}: " +
                     ("one LambdaLR per optimiser (Train.lr_scheduler_per_optimizer: true) -- NOT what the reference does."
                      if per_optimizer else
                      f"as in the reference (models/base_model_abc.py:63-64), every LambdaLR is attached to the LAST optimiser ({names[-1]}): "
                      f"{', '.join(names[:-1])} keep their initial learning rate. Train.lr_scheduler_per_optimizer: true decays all of them."))
            self.lr_schedulers = [torch.optim.lr_scheduler.LambdaLR(getattr(self, name if per_optimizer else names[-1]), schedule) for name in names]
            if getattr(args, "start_epoch", 0) > 0:
                self._resume(model_path, device)
            else:
                for net_name in [n for names in self.optimizer_mapping.values() for n in names]:
                    m: nn.Module = getattr(self, net_name)
                    activation = "relu" if "resnet" in m._get_name().lower() else "leaky_relu"
                    init_weights(m, init_type="kaiming", nonlinearity=activation)
                    _log(f"Initialized {net_name} network weights using He initialization ({activation}).")
            if _dist_on():                                     # replicas start from rank 0's weights
                for p in self.parameters():
                    dist.broadcast(p.data, src=0)
                self._after_weight_surgery()
            # single process: gradients are handed over by autograd without an accumulation pass (grad = None before backward);
            # data-parallel: they accumulate into the arena views, which IS the all-reduce bucket
            if _dist_on() or os.environ.get("OCTA_GRAD_ARENA") == "1":
                self._arenas = {name: GradArena(itertools.chain(*[n.parameters() for n in self._nets_of(net_names)]))
                                for name, net_names in self.optimizer_mapping.items()}
        else:
            # `General.inference` names the network: `model`, `segmentor` / `generator`, or the aliases `S` / `G` the GAN-seg
            # config ships. train.py writes `<epoch>_<network>_model.pth` (utils/visualizer.py:225-238), so the alias is resolved
            # BEFORE the path is built (the reference builds `<epoch>_G_model.pth`, which its own trainer never writes); the alias
            # spelling and the legacy `<epoch>_model.pth` are still accepted.
            alias = config["General"].get("inference")
            net = {"S": "segmentor", "G": "generator"}.get(alias, alias)
            cands = [model_path.replace("model.pth", f"{n}_model.pth") for n in dict.fromkeys((net, alias)) if n]
            if net in (None, "model"):       # only the un-named / `model` network may fall back to the generic files: for `S` / `G` another
                cands += [model_path.replace("model.pth", "model_model.pth")]      # network's checkpoint would be picked silently otherwise
            cands += [model_path]            # the reference's legacy single file `<epoch>_model.pth` (base_model_abc.py:104-107)
            checkpoint_path = next((c for c in cands if os.path.exists(c)), None)
            if checkpoint_path is None:
                raise FileNotFoundError(f"no checkpoint for inference={alias!r}; looked for {cands}")
            checkpoint = load_checkpoint_file(checkpoint_path, device)
            which = config["General"].get("inference") or "model"
            which = {"S": "segmentor", "G": "generator"}.get(which, which)
            config["General"]["inference"] = which
            assert hasattr(self, which), f"Inference mode {which} not implemented."
            getattr(self, which).load_state_dict(checkpoint["model"])
            self._after_weight_surgery()
            _log(f"Loaded network weights {which} from epoch {checkpoint['epoch']}.")

    def _resume(self, model_path, device):
        for optimizer_name, net_names in self.optimizer_mapping.items():
            checkpoint = None
            if len(net_names) == 0:
                checkpoint = load_checkpoint_file(model_path, device)
                self.load_state_dict(checkpoint["model"])
            for net_name in net_names:
                checkpoint = load_checkpoint_file(model_path.replace("model.pth", f"{net_name}_model.pth"), device)
                getattr(self, net_name).load_state_dict(checkpoint["model"])
            optimizer: torch.optim.Optimizer = getattr(self, optimizer_name)
            if checkpoint.get("optimizer") is not None:
                optimizer.load_state_dict(checkpoint["optimizer"])
            else:
                for cand in (model_path.replace("model.pth", f"{optimizer_name}.pth"), model_path.replace("model.pth", f"{optimizer_name}_model.pth")):
                    if os.path.exists(cand):
                        optimizer.load_state_dict(load_checkpoint_file(cand, device)["optimizer"])
                        break
                else:
                    raise FileNotFoundError(f"no optimizer state for {optimizer_name} next to {model_path}")
            _log(f"Loaded all network weights from epoch {checkpoint['epoch']}.")
        self._after_weight_surgery()

    def _after_weight_surgery(self):
        """Weights were written through .data / load_state_dict: packed bf16 copies of the MFMA path are stale."""
        from . import mfma_conv
        mfma_conv.invalidate_all_pack_plans(self)

    # ---- the step ------------------------------------------------------------------------------------------------
    def autocast(self):
        dev = next(self.parameters()).device
        return torch.autocast(device_type=dev.type, dtype=torch.bfloat16, enabled=self.amp and dev.type == "cuda")

    def backward_scope(self):
        """Scope of a `loss.backward()` of the step: the 3x3 weight-gradient launches add their result straight to the parameters' `.grad`
        (views of the flat gradient arena) instead of handing autograd a tensor per layer (models/mfma_conv.py: direct_weight_grads)."""
        from . import mfma_conv
        return mfma_conv.direct_weight_grads(next(self.parameters()).device)

    def zero_grads(self, optimizer_name):
        if optimizer_name in self._arenas:
            self._arenas[optimizer_name].zero()
        else:
            getattr(self, optimizer_name).zero_grad(set_to_none=True)

    def exchange_gradients(self, *optimizer_names):
        """Data-parallel mean of the gradients of the named optimisers: one all-reduce per arena."""
        for name in optimizer_names:
            if name in self._arenas:
                self._arenas[name].all_reduce_mean()

    @abstractmethod
    def inference(self, mini_batch: Dict[str, Any], post_transformations: Dict[str, Callable], device: torch.device = "cpu",
                  phase: Phase = Phase.TEST) -> Tuple[Output, Dict[str, torch.Tensor]]:
        raise NotImplementedError()

    @abstractmethod
    def forward(self, input: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError()

    def perform_training_step(self, mini_batch: Dict[str, Any], scaler, post_transformations: Dict[str, Callable],
                              device: torch.device = "cpu") -> Tuple[Output, Dict[str, float]]:
        self.zero_grads("optimizer")
        with self.autocast():
            outputs, losses = self.inference(mini_batch, post_transformations, device, phase=Phase.TRAIN)
            loss = sum(list(losses.values()))
        with self.backward_scope():
            loss.backward()
        self.exchange_gradients("optimizer")
        self.optimizer.step()
        return outputs, LossValues(losses)

    def compute_metric(self, outputs: Output, metrics) -> None:
        metrics(y_pred=outputs["prediction"], y=outputs["label"])


class LossValues(dict):
    """Losses of a training step. The reference returns python floats (`v.item()` per loss: one device sync each,
    base_model_abc.py:166); here the values stay device tensors until someone reads them -- `float(v)`, `v.item()`,
    arithmetic and formatting all work on 0-dim tensors -- so a step queues no sync of its own. `as_floats()` gives
    the reference's dict in one transfer."""

    def as_floats(self):
        keys = list(self.keys())
        vals = torch.stack([torch.as_tensor(self[k]).detach().float().reshape(()) for k in keys]).cpu().tolist()
        return dict(zip(keys, vals))
