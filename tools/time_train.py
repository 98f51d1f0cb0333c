"""Device timing of the DynUNet-S training step at 1x1216x1216 (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
CFG = {"General": {"amp": True, "model": {"name": "DynUNet", "spatial_dims": 2, "in_channels": 1, "out_channels": 1,
                                           "kernel_size": [3, 3, 3, 3, 3], "strides": [1, 2, 2, 2, 1],
                                           "upsample_kernel_size": [1, 2, 2, 2, 1]}},
       "Train": {"lr": 1e-4, "loss": "DiceBCELoss", "epochs": 30, "epochs_decay": 10}}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
res = int(sys.argv[2]) if len(sys.argv) > 2 else 1216
cl = (sys.argv[3] == "channels_last") if len(sys.argv) > 3 else False      # default: the bench's configuration (plain parameters: the pack plan covers them)
tr = SegmentationTrainer(CFG, "cuda", channels_last=cl)
x = torch.rand(B, 1, res, res, device="cuda"); y = (torch.rand(B, 1, res, res, device="cuda") > 0.8).float()
for i in range(3):
    tr.perform_training_step({"image": x, "label": y})
torch.cuda.synchronize(); t = time.time(); n = 10
for i in range(n):
    out, losses = tr.perform_training_step({"image": x, "label": y})
torch.cuda.synchronize(); dt = time.time() - t
print(f"DynUNet train B={B} {res}x{res} channels_last={cl}: {dt/n*1e3:.1f} ms/step -> {B*n/dt:.1f} imgs/s, loss {float(list(losses.values())[0]):.4f}, "
      f"~{2.0*B*n/dt:.1f} TFLOP/s (2.0 TFLOP per image), mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
