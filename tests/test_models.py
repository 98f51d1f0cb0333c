"""CPU: DynUNet restatement (shape, parameter count, MONAI-style state_dict keys), losses, and the
data-parallel gradient exchange over gloo with two processes."""
import os
import socket

import numpy as np
import pytest
import torch

from octa_autosegmentation_amd.models import losses, networks

CFG = {"General": {"amp": True, "model": {"name": "DynUNet", "spatial_dims": 2, "in_channels": 1, "out_channels": 1,
                                           "kernel_size": [3, 3, 3, 3, 3], "strides": [1, 2, 2, 2, 1],
                                           "upsample_kernel_size": [1, 2, 2, 2, 1]}},
       "Train": {"lr": 1e-4, "loss": "DiceBCELoss", "epochs": 30, "epochs_decay": 10}}


def test_dynunet_shape_params_and_keys():
    kw = dict(CFG["General"]["model"]); kw.pop("name")
    net = networks.DynUNet(**kw)
    assert sum(p.numel() for p in net.parameters()) == 7_368_769
    x = torch.randn(2, 1, 64, 64)
    assert net(x).shape == (2, 1, 64, 64)
    sd = net.state_dict()
    for k in ("input_block.conv1.conv.weight", "downsamples.2.norm2.bias", "bottleneck.conv2.conv.weight",
              "upsamples.0.transp_conv.conv.weight", "upsamples.3.conv_block.conv1.conv.weight",
              "output_block.conv.conv.bias", "skip_layers.downsample.conv1.conv.weight",
              "skip_layers.next_layer.next_layer.next_layer.next_layer.conv1.conv.weight",
              "skip_layers.upsample.transp_conv.conv.weight"):
        assert k in sd, k
    assert sd["upsamples.0.transp_conv.conv.weight"].shape == (512, 256, 1, 1)
    assert sd["upsamples.3.transp_conv.conv.weight"].shape == (64, 32, 2, 2)
    net2 = networks.DynUNet(**kw)
    net2.load_state_dict(sd)            # aliases are ignored on load
    assert torch.equal(net2(x), net(x))


def test_dice_bce_formula():
    torch.manual_seed(0)
    logits, y = torch.randn(3, 1, 16, 16), (torch.rand(3, 1, 16, 16) > 0.7).float()
    p = torch.sigmoid(logits)
    dice = torch.mean(torch.stack([1 - (2 * (p[b] * y[b]).sum() + 1e-5) / (p[b].sum() + y[b].sum() + 1e-5) for b in range(3)]))
    want = (dice + torch.nn.functional.binary_cross_entropy_with_logits(logits, y)) / 2
    assert torch.allclose(losses.DiceBCELoss(True)(logits, y), want, atol=1e-7)


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                      # different init per rank: must be overwritten by rank 0's
    tr = SegmentationTrainer(CFG, "cpu", channels_last=False)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(2, 1, 32, 32, generator=g); y = (torch.rand(2, 1, 32, 32, generator=g) > 0.6).float()
    tr.perform_training_step({"image": x[rank:rank + 1], "label": y[rank:rank + 1]})
    if rank == 0:
        torch.save({k: v.clone() for k, v in tr.model.state_dict().items() if not k.startswith("skip_layers")}, out)
    dist.destroy_process_group()


def test_two_rank_data_parallel_step_equals_single_process_batch(tmp_path):
    import torch.multiprocessing as mp
    from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(100)
    ref = SegmentationTrainer(CFG, "cpu", channels_last=False)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(2, 1, 32, 32, generator=g); y = (torch.rand(2, 1, 32, 32, generator=g) > 0.6).float()
    # mean over ranks of per-rank (batch 1) gradients == gradient of the mean of the two per-sample losses
    ref.optimizer.zero_grad()
    l = sum(ref.loss_function(ref.model(x[i:i + 1]), y[i:i + 1]) for i in range(2)) / 2
    l.backward(); ref.optimizer.step()
    for k, v in ref.model.state_dict().items():
        if k.startswith("skip_layers"):
            continue
        assert torch.allclose(got[k], v, atol=1e-6), k
