"""Device timing of the GAN-seg training step (BASELINE configs[3], one GPU): G, D at 304^2, S at 1216^2 (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octa_autosegmentation_amd.models.gan_seg_trainer import GanSegTrainer
S = {"name": "DynUNet", "spatial_dims": 2, "in_channels": 1, "out_channels": 1, "kernel_size": [3, 3, 3, 3, 3], "strides": [1, 2, 2, 2, 1],
     "upsample_kernel_size": [1, 2, 2, 2, 1]}
cfg = {"General": {"amp": True, "model": {"name": "GanSegModel", "model_g": {"name": "resnetGenerator9"}, "model_d": {"name": "patchGAN70x70"},
                                            "model_s": S, "upshape": (1216, 1216)}},
       "Train": {"lr": 2e-4, "loss_dg": "LSGANLoss", "loss_s": "DiceBCELoss"}}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tr = GanSegTrainer(cfg, "cuda")
batch = {"real_A": torch.rand(B, 1, 304, 304, device="cuda"), "real_B": torch.rand(B, 1, 304, 304, device="cuda"),
         "real_A_seg": (torch.rand(B, 1, 1216, 1216, device="cuda") > 0.8).float()}
for _ in range(3):
    tr.perform_training_step(batch)
torch.cuda.synchronize(); t = time.time(); n = 10
for _ in range(n):
    out, losses = tr.perform_training_step(batch)
torch.cuda.synchronize(); dt = time.time() - t
print(f"GAN-seg step B={B}: {dt/n*1e3:.1f} ms/step -> {B*n/dt:.1f} imgs/s, losses " + ", ".join(f"{k}={float(v):.3f}" for k, v in losses.items()))
