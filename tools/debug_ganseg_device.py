"""Diagnostic (round 3): the GAN-seg fixture on cpu / cuda-fp32 / cuda-bf16 -- per-step losses, per-network gradient norms after
EACH step and the largest per-parameter gradient deviations of the first step (before any optimiser update)."""
import os
import sys
from argparse import Namespace
from copy import deepcopy

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.make_golden_ganseg import S_CFG, TRAIN, WEIGHTS, batch  # noqa: E402
from octa_autosegmentation_amd.models import networks  # noqa: E402
from octa_autosegmentation_amd.models.model import define_model  # noqa: E402
from octa_autosegmentation_amd.utils.enums import Phase  # noqa: E402


def build(device, amp, idt, variant="he_"):
    config = {"General": {"device": device, "amp": amp, "model": {"name": "GanSegModel", "model_g": {"name": "resnetGenerator9"},
                                                                   "model_d": {"name": "patchGAN70x70"}, "model_s": dict(S_CFG),
                                                                   "compute_identity": idt, "compute_identity_seg": True, "upshape": (64, 64)}},
              "Train": dict(TRAIN), "Output": {"save_dir": "/tmp"}}
    torch.manual_seed(0)
    model = define_model(deepcopy(config), Phase.TRAIN)
    model.initialize_model_and_optimizer(None, networks.init_weights, config, Namespace(start_epoch=0, epoch="latest"), None, Phase.TRAIN)
    for salt, name in ((0, "generator"), (100, "discriminator"), (200, "segmentor")):
        WEIGHTS[variant](getattr(model, name), salt)
    model._after_weight_surgery()
    model.train()
    return model


def grads(model):
    out = {}
    for net in ("generator", "discriminator", "segmentor"):
        for n, p in getattr(model, net).named_parameters():
            out[f"{net}.{n}"] = None if p.grad is None else p.grad.detach().double().cpu().clone()
    return out


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "ganseg_golden.npz"))
    ident = {"prediction": lambda t: t, "label": lambda t: t}
    keys = ("S", "D_fake", "D_real", "G", "G_idt", "S_idt")
    tag, idt = "he_idt0", False
    print("golden losses", g[f"{tag}_losses"], "gnorm", g[f"{tag}_grad_norms"])
    ref = None
    for device, amp in (("cpu", False), ("cuda", False), ("cuda", True)):
        m = build(device, amp, idt)
        for step in range(2):
            _, l = m.perform_training_step(batch(), None, ident, device)
            gr = grads(m)
            norms = {net: float(np.sqrt(sum(float((v ** 2).sum()) for k, v in gr.items() if k.startswith(net) and v is not None))) for net in ("generator", "discriminator", "segmentor")}
            print(device, "amp" if amp else "fp32", "step", step, [round(float(l[k]), 6) for k in keys], norms, flush=True)
            if step == 0:
                if ref is None:
                    ref = gr
                else:
                    worst = []
                    for k, v in gr.items():
                        r = ref[k]
                        if v is None or r is None:
                            if (v is None) != (r is None):
                                worst.append((float("inf"), k, "None mismatch"))
                            continue
                        d = float((v - r).norm()) / max(float(r.norm()), 1e-30)
                        worst.append((d, k, f"|ref| {float(r.norm()):.3e}"))
                    worst.sort(reverse=True)
                    for w in worst[:12]:
                        print("    ", w)


if __name__ == "__main__":
    main()
