"""GAN-seg training step replayed from a HIP graph (torch.cuda.CUDAGraph) against the eager step: how much of the step is launch
overhead / dependency gaps (development aid; the optimisers are rebuilt with capturable=True for the capture)."""
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
for name in tr.impl.optimizer_mapping:                      # Adam inside a captured region needs device-side step counters
    opt = getattr(tr.impl, name)
    for g in opt.param_groups:
        g["capturable"] = True
batch = {"real_A": torch.rand(B, 1, 304, 304, device="cuda"), "real_B": torch.rand(B, 1, 304, 304, device="cuda"),
         "real_A_seg": (torch.rand(B, 1, 1216, 1216, device="cuda") > 0.8).float()}
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(4):
        tr.perform_training_step(batch)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize(); t = time.time(); n = 10
for _ in range(n):
    tr.perform_training_step(batch)
torch.cuda.synchronize(); eager = (time.time() - t) / n
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out, losses = tr.perform_training_step(batch)
torch.cuda.synchronize(); t = time.time()
for _ in range(n):
    g.replay()
torch.cuda.synchronize(); dt = (time.time() - t) / n
print(f"GAN-seg step B={B}: eager {eager*1e3:.1f} ms, graph replay {dt*1e3:.1f} ms; losses " + ", ".join(f"{k}={float(v):.3f}" for k, v in losses.items()))
