"""Segmentation training step of the reference (LambdaModel + BaseModelABC.perform_training_step,
models/lambda_model.py:13-48, models/base_model_abc.py:25-92,152-167) for one process per GPU:
Adam(lr, betas (0.5, 0.999)), LambdaLR (constant, then linear decay over `epochs_decay`), He init,
autocast forward + DiceBCE loss, backward, optimizer step. Mixed precision is bf16 (no GradScaler
needed; the reference's fp16 + GradScaler is an artefact of its CUDA target). With
torch.distributed initialised (backend "nccl" = RCCL over xGMI), gradients of all ranks are averaged
with ONE all-reduce over a flat bucket per step (7.4 M parameters = 29.5 MB fp32)."""
import os

# MIOpen's exhaustive find mode benchmarks every solver (incl. naive reference kernels) on first use:
# minutes per process on a fresh box. The fast heuristic mode starts in seconds (set before torch loads MIOpen).
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
# In FAST mode MIOpen answers the backward-data convolutions of this network with GEMM + Col2Im2dU (20 % of the
# step in profiles/r01_train_kernel_stats.csv); without the GEMM solvers it picks its implicit-GEMM kernels
# (53 -> 41 ms per step at B=4, 1216x1216).
os.environ.setdefault("MIOPEN_DEBUG_CONV_GEMM", "0")
# ... and its asm NHWC data-gradient kernel is launched without the workspace it asks for (faults; see gan_seg_trainer.py)
os.environ.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC", "0")

import torch
import torch.distributed as dist

from .losses import get_loss_function_by_name
from .networks import MODEL_DICT, init_weights


class SegmentationTrainer:
    optimizer_mapping = {"optimizer": ["model"]}

    def __init__(self, config, device, channels_last=False):
        kw = dict(config["General"]["model"])
        name = kw.pop("name")
        self.device = torch.device(device)
        self.model = MODEL_DICT[name](**kw).to(self.device)
        if channels_last:
            self.model = self.model.to(memory_format=torch.channels_last)
        self.channels_last = channels_last
        init_weights(self.model, init_type="kaiming", nonlinearity="leaky_relu")
        tr = config["Train"]
        self.loss_name = tr.get("loss", "DiceBCELoss")
        self.loss_function = get_loss_function_by_name(self.loss_name, config)
        self.optimizer = torch.optim.Adam(self.model.parameters(), lr=tr["lr"], betas=(0.5, 0.999),
                                          weight_decay=tr.get("weight_decay", 0))
        max_epochs, decay = tr["epochs"], tr.get("epochs_decay", 0)
        self.lr_schedulers = [torch.optim.lr_scheduler.LambdaLR(
            self.optimizer, lambda step: 1 if step < (max_epochs - decay) else (max_epochs - step) * (1 / max(1, decay)))]
        self.amp = bool(config["General"].get("amp", True)) and self.device.type == "cuda"
        self._flat = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            self._broadcast_parameters()

    def _broadcast_parameters(self):
        for p in self.model.parameters():
            dist.broadcast(p.data, src=0)

    def _allreduce_gradients(self):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
        grads = [p.grad for p in self.model.parameters() if p.grad is not None]
        n = sum(g.numel() for g in grads)
        if self._flat is None or self._flat.numel() != n:
            self._flat = torch.empty(n, dtype=torch.float32, device=grads[0].device)
        off = 0
        for g in grads:
            self._flat[off:off + g.numel()].copy_(g.reshape(-1))
            off += g.numel()
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM)
        self._flat.div_(dist.get_world_size())
        off = 0
        for g in grads:
            g.copy_(self._flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def forward(self, x):
        return self.model(x)

    def perform_training_step(self, mini_batch, scaler=None, post_transformations=None, device=None):
        x = mini_batch["image"].to(self.device, non_blocking=True)
        y = mini_batch["label"].to(self.device, non_blocking=True)
        if self.channels_last:
            x = x.contiguous(memory_format=torch.channels_last)
        self.optimizer.zero_grad(set_to_none=True)
        with torch.autocast(device_type=self.device.type, dtype=torch.bfloat16, enabled=self.amp):
            pred = self.model(x)
            loss = self.loss_function(pred.float(), y.float())
        loss.backward()
        self._allreduce_gradients()
        self.optimizer.step()
        return {"prediction": pred}, {self.loss_name: loss}
