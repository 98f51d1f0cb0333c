"""tools/make_golden_ganseg.py -- generates tests/golden/ganseg_golden.npz: the reference's OWN joint G / D / S update
(models/gan_seg_model.py:110-173 driven through BaseModelABC.initialize_model_and_optimizer, base_model_abc.py:25-92) recorded on
seeded CPU inputs, so that a sign error or a wrong detach in this repository's GanSegModel fails a test (row a21).

Runs ONLY in the build container: imports /root/reference (read-only). What the reference imports but the image lacks is
stood in for as SURVEY.md 8c describes:
  * monai            -> MagicMock, except `monai.data.decollate_batch` (list of leading-dimension slices) and
                        `monai.losses.DiceLoss`, for which this repository's restatement of MONAI's documented formula is
                        used (Dice itself stays MONAI-unpinned; this fixture pins the update logic around it);
  * the segmentor    -> this repository's DynUNet (MONAI's is absent) registered under MODEL_DICT["DynUNet"]; generator and
                        discriminator are the reference's own classes.
All parameters are set from the closed form `fill` (in state_dict order), inputs from `image`, so the test rebuilds them
without any weights travelling. Recorded per variant (compute_identity on / off): the six losses of two consecutive steps,
gradient norms of G / D / S after the second step, float64 checksums of every network's parameters after the two steps.
"""
import argparse
import os
import sys
import warnings
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden", "ganseg_golden.npz")
sys.path.insert(0, ROOT)
from tools.make_golden_networks import fill, image  # noqa: E402

S_CFG = {"name": "DynUNet", "spatial_dims": 2, "in_channels": 1, "out_channels": 1, "kernel_size": [3, 3, 3, 3, 3], "strides": [1, 2, 2, 2, 1],
         "upsample_kernel_size": [1, 2, 2, 2, 1]}
TRAIN = {"lr": 2e-4, "epochs": 100, "epochs_decay": 0, "loss_dg": "LSGANLoss", "loss_s": "DiceBCELoss", "batch_size": 2}


def formula_weights(net, salt):
    sd = net.state_dict()
    for k, (name, t) in enumerate(sd.items()):
        if name.endswith("filt") or name.startswith("skip_layers"):
            continue
        sd[name] = fill(tuple(t.shape), k + salt)
    net.load_state_dict(sd)


def he_weights(net, salt):
    """Well-conditioned closed-form parameters (round 3): per tensor (state_dict order, index k) numpy's legacy
    RandomState(7000 + salt + k) -- a fixed algorithm, so the test rebuilds the same values on any machine; conv weights
    N(0, 2 / fan_in) (He), InstanceNorm scales 1 + 0.1 N(0, 1), biases 0.01 N(0, 1). The `fill` ramp above makes some
    InstanceNorm inputs nearly constant (gradient norms of 1e6 in the generator), which amplifies summation-order differences
    between devices by orders of magnitude; these parameters do not."""
    sd = net.state_dict()
    for k, (name, t) in enumerate(sd.items()):
        if name.endswith("filt") or name.startswith("skip_layers"):
            continue
        r = np.random.RandomState(7000 + salt + k).standard_normal(tuple(t.shape))
        if t.dim() > 1:
            v = r * np.sqrt(2.0 / max(int(np.prod(t.shape[1:])), 1))
        elif name.endswith("bias"):
            v = 0.01 * r
        else:
            v = 1.0 + 0.1 * r
        sd[name] = torch.from_numpy(v.astype(np.float32))
    net.load_state_dict(sd)


WEIGHTS = {"": formula_weights, "he_": he_weights}


def batch():
    real_A = image((2, 1, 32, 32), 4)
    real_B = image((2, 1, 32, 32), 5).flip(-1)
    seg = (image((2, 1, 64, 64), 6) > 0.55).float()
    return {"real_A": real_A, "real_B": real_B, "real_A_seg": seg}


def checksums(model):
    out = []
    for name in ("generator", "discriminator", "segmentor"):
        ps = [p.detach().double() for n, p in getattr(model, name).named_parameters()]
        out.append([float(sum(p.sum() for p in ps)), float(sum(p.abs().sum() for p in ps))])
    return np.array(out)


def grad_norms(model):
    return np.array([float(torch.sqrt(sum((p.grad.double() ** 2).sum() for p in getattr(model, n).parameters() if p.grad is not None)))
                     for n in ("generator", "discriminator", "segmentor")])


def run(model, steps=2):
    from torch.amp import GradScaler
    scaler = GradScaler("cpu", enabled=False)
    ident = {"prediction": lambda t: t, "label": lambda t: t}
    losses, norms = [], []
    for _ in range(steps):
        _, l = model.perform_training_step(batch(), scaler, ident, "cpu")
        losses.append([float(l[k]) for k in ("S", "D_fake", "D_real", "G", "G_idt", "S_idt")])
        norms.append(grad_norms(model))
    return np.array(losses), np.array(norms), checksums(model)


def main():
    warnings.filterwarnings("ignore")
    sys.path.insert(0, "/root/reference")
    from octa_autosegmentation_amd.models import losses as our_losses
    from octa_autosegmentation_amd.models.networks import DynUNet
    monai = MagicMock()
    monai.data.decollate_batch = lambda t: [t[i] for i in range(t.shape[0])]

    class DiceLoss(torch.nn.Module):
        def __init__(self, sigmoid=False, **kw):
            super().__init__()
            self.f = our_losses.DiceLoss(sigmoid=sigmoid)

        def forward(self, y_pred, y):
            return self.f(y_pred, y)

    monai.losses.DiceLoss = DiceLoss
    for m in ["monai", "monai.data", "monai.losses", "monai.networks", "monai.networks.nets", "monai.networks.blocks", "monai.networks.layers",
              "monai.metrics", "monai.transforms", "monai.utils", "monai.config", "skimage", "skimage.filters", "skimage.morphology", "nibabel",
              "prettytable", "natsort", "torchvision", "torchvision.models", "torchvision.transforms", "torchvision.transforms.functional", "matplotlib",
              "matplotlib.pyplot", "rich", "rich.progress", "rich.live", "rich.spinner", "typing_extensions_none",
              "rich.console", "models.oof", "models.frangi", "models.skrgan", "models.nice_gan", "models.cycle_gan", "models.cut", "models.negcut",
              "models.dclgan", "models.noise_model", "utils.cldice"]:
        sys.modules.setdefault(m, monai if m == "monai" else MagicMock())
    sys.modules["monai.data"] = monai.data
    sys.modules["monai.losses"] = monai.losses
    import importlib
    ref_nets = importlib.import_module("models.networks")
    ref_gs = importlib.import_module("models.gan_seg_model")
    from utils.enums import Phase
    MODEL_DICT = {"resnetGenerator9": ref_nets.resnetGenerator9, "patchGAN70x70": ref_nets.patchGAN70x70, "DynUNet": DynUNet}
    out = {}
    for tag, idt in (("idt0", False), ("idt1", True), ("he_idt0", False), ("he_idt1", True)):
        set_weights = WEIGHTS["he_" if tag.startswith("he_") else ""]
        torch.manual_seed(0)
        model = ref_gs.GanSegModel(MODEL_DICT, {"name": "resnetGenerator9"}, {"name": "patchGAN70x70"}, dict(S_CFG), compute_identity=idt,
                                   compute_identity_seg=True, phase=Phase.TRAIN, upshape=(64, 64))
        config = {"General": {"device": "cpu", "amp": False}, "Train": dict(TRAIN), "Output": {"save_dir": "/tmp"}}
        model.initialize_model_and_optimizer(None, ref_nets.init_weights, config, argparse.Namespace(start_epoch=0, epoch="latest"), None, Phase.TRAIN)
        for salt, name in ((0, "generator"), (100, "discriminator"), (200, "segmentor")):
            set_weights(getattr(model, name), salt)
        model.train()
        losses, gnorm, sums = run(model)
        # `_grad_norms`: after the second step (round 2's entry); `_grad_norms_steps`: after each step -- the first row is a pure
        # backward pass (no optimiser update behind it), the sharp pin of detach / sign errors on reduced-precision paths
        out[f"{tag}_losses"], out[f"{tag}_grad_norms"], out[f"{tag}_grad_norms_steps"], out[f"{tag}_param_sums"] = losses, gnorm[-1], gnorm, sums
        print(tag, losses, gnorm, sums, sep="\n")
    np.savez_compressed(OUT, **out)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
