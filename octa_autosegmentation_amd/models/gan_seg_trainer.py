"""Programmatic front end of the joint GAN + segmentation step (models/gan_seg_model.py) for bench.py /
train_synthetic.py / tests: builds GanSegModel through define_model from a config dict and drives its training step on
device-resident batches."""
import os
from argparse import Namespace
from copy import deepcopy

import torch

from ..utils.enums import Phase
from .model import define_model
from .networks import init_weights
from .segmentation_trainer import _complete


class GanSegTrainer:
    def __init__(self, config, device, upshape=(1216, 1216), args=None):
        """args: the command line namespace of train.py (`start_epoch` > 0 resumes from `Output.save_dir`/checkpoints/`epoch`_*); None = a fresh model."""
        self.device = torch.device(device)
        self.config = _complete(config, self.device)
        self.config["General"]["model"].setdefault("upshape", tuple(upshape))
        tr = self.config.setdefault("Train", {})
        tr.setdefault("loss_dg", "LSGANLoss"); tr.setdefault("loss_s", "DiceBCELoss")
        tr.setdefault("epochs", 100); tr.setdefault("epochs_decay", 0)
        self.impl = define_model(deepcopy(self.config), Phase.TRAIN)
        self.impl.initialize_model_and_optimizer(None, init_weights, self.config, args or Namespace(start_epoch=0, epoch="latest"), None, Phase.TRAIN)
        self.impl.train()

    def __getattr__(self, name):            # generator, discriminator, segmentor, optimizer_G/D/S, lr_schedulers, ...
        return getattr(self.__dict__["impl"], name)

    def perform_training_step(self, mini_batch, scaler=None, post_transformations=None, device=None):
        outputs, losses = self.impl.perform_training_step(mini_batch, scaler, post_transformations, self.device)
        return {**outputs, "prediction": outputs["prediction"][0].unsqueeze(0), "label": outputs["label"][0].unsqueeze(0)}, losses
