"""Programmatic front end of the segmentation training step for bench.py / train_synthetic.py / tests: builds the
ModelInterface model the entry points use (define_model -> LambdaModel(DynUNet), models/lambda_model.py) from a config dict
and drives its `perform_training_step` on tensors that are already batched on the device. All behaviour (Adam + LambdaLR,
He init, bf16 autocast, flat gradient arena + one RCCL all-reduce per step) lives in models/base_model_abc.py."""
import os
from argparse import Namespace
from copy import deepcopy

import torch

from ..utils.enums import Phase
from .model import define_model
from .networks import init_weights

IDENTITY_POST = {"prediction": lambda t: t, "label": lambda t: t}


def _complete(config, device):
    cfg = deepcopy(config)
    cfg.setdefault("General", {})["device"] = str(device)
    cfg.setdefault("Output", {}).setdefault("save_dir", ".")
    return cfg


class SegmentationTrainer:
    def __init__(self, config, device, channels_last=False):
        self.device = torch.device(device)
        self.config = _complete(config, self.device)
        self.impl = define_model(deepcopy(self.config), Phase.TRAIN)
        if channels_last:
            self.impl.model = self.impl.model.to(memory_format=torch.channels_last)
        self.channels_last = channels_last
        self.impl.initialize_model_and_optimizer(None, init_weights, self.config, Namespace(start_epoch=0, epoch="latest"), None, Phase.TRAIN)
        self.impl.train()

    def __getattr__(self, name):            # model, optimizer, lr_schedulers, loss_name, loss_function, optimizer_mapping, ...
        return getattr(self.__dict__["impl"], name)

    def forward(self, x):
        return self.impl(x)

    def perform_training_step(self, mini_batch, scaler=None, post_transformations=None, device=None):
        x = mini_batch["image"]
        if self.channels_last:
            x = x.to(self.device).contiguous(memory_format=torch.channels_last)
        outputs, losses = self.impl.perform_training_step({**mini_batch, "image": x}, scaler, post_transformations or IDENTITY_POST, self.device)
        return {"prediction": outputs["prediction"][0].unsqueeze(0)}, losses
