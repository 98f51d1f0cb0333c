"""Which ATen operators (the tiny launches between the hand-written kernels) one U-Net training step runs, and from where (development aid)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
CFG = {"General": {"amp": True, "model": {"name": "DynUNet", "spatial_dims": 2, "in_channels": 1, "out_channels": 1,
                                           "kernel_size": [3, 3, 3, 3, 3], "strides": [1, 2, 2, 2, 1], "upsample_kernel_size": [1, 2, 2, 2, 1]}},
       "Train": {"lr": 1e-4, "loss": "DiceBCELoss", "epochs": 30, "epochs_decay": 10}}
res = int(sys.argv[1]) if len(sys.argv) > 1 else 608
tr = SegmentationTrainer(CFG, "cuda")
x = torch.rand(4, 1, res, res, device="cuda"); y = (torch.rand(4, 1, res, res, device="cuda") > 0.8).float()
for _ in range(3):
    tr.perform_training_step({"image": x, "label": y})
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    tr.perform_training_step({"image": x, "label": y})
    torch.cuda.synchronize()
ops = collections.Counter()
where = collections.defaultdict(collections.Counter)
LEAF = {"aten::flip", "aten::copy_", "aten::fill_", "aten::zero_", "aten::add_", "aten::mul", "aten::add", "aten::sub", "aten::div", "aten::mean", "aten::sum",
        "aten::_to_copy", "aten::clone", "aten::contiguous", "aten::rsub", "aten::empty", "aten::zeros", "aten::cat", "aten::_foreach_add_", "aten::mul_"}
for e in prof.events():
    if e.name in LEAF:
        ops[e.name] += 1
        frames = [f for f in (e.stack or []) if "octa_autosegmentation_amd" in f or "torch/optim" in f or "tools/" in f]
        where[e.name][(frames[0].split("octa_autosegmentation_amd/")[-1] if frames else "?")[:120]] += 1
for name, n in ops.most_common():
    print(f"{n:4d} {name}")
    for w, c in where[name].most_common(8):
        print(f"        {c:3d} {w}")
