"""What can be pinned of the MONAI boundary without MONAI (rows a18 / a23 stay "MONAI-unpinned" for NUMERICS):
the checkpoint contract of DynUNet -- state-dict keys and shapes derived from MONAI's published module structure by
tools/make_monai_manifest.py, independently of this repository's network -- and DiceLoss(sigmoid=True) on hand-computed cases."""
import json
import math
import os

import torch

from octa_autosegmentation_amd.models import losses, networks

M = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "monai_dynunet_S_manifest.json")))


def _net():
    c = M["ctor"]
    return networks.DynUNet(c["spatial_dims"], c["in_channels"], c["out_channels"], c["kernel_size"], c["strides"], c["upsample_kernel_size"])


def test_state_dict_has_monais_keys_shapes_and_order():
    sd = _net().state_dict()
    want = [(k, tuple(s)) for k, s in M["keys"]]
    got = [(k, tuple(v.shape)) for k, v in sd.items()]
    assert dict(got) == dict(want)                                   # same names, same shapes (skip_layers.* aliases included)
    own = [k for k, _ in got if not k.startswith("skip_layers.")]
    assert own == [k for k, _ in want if not k.startswith("skip_layers.")]      # registration order of the real modules
    assert sum(p.numel() for p in _net().parameters()) == M["n_parameters"] == 7368769


def test_a_monai_shaped_checkpoint_loads_strictly_and_lands():
    """A state dict with EXACTLY MONAI's keys (what torch.save(model.state_dict()) of the reference's trainer holds, shared tensors listed
    under both names) loads with strict=True, every tensor lands in the parameter MONAI's name points at, and what this model saves
    loads back into a fresh model."""
    g = torch.Generator().manual_seed(0)
    own = {k: torch.randn(tuple(s), generator=g) for k, s in M["keys"] if not k.startswith("skip_layers.")}
    net = _net()
    alias = {dst: src for src, dst in net._alias_prefixes().items()}            # "skip_layers...." prefix -> real module prefix
    sd = {}
    for k, s in M["keys"]:
        if k.startswith("skip_layers."):
            pre = max((p for p in alias if k.startswith(p)), key=len)
            sd[k] = own[alias[pre] + k[len(pre):]]
        else:
            sd[k] = own[k]
    missing, unexpected = net.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    for k, v in own.items():
        assert torch.equal(dict(net.named_parameters())[k], v), k
    again = _net()
    again.load_state_dict(net.state_dict(), strict=True)
    assert all(torch.equal(a, b) for a, b in zip(net.parameters(), again.parameters()))
    x = torch.rand(1, 1, 32, 32)
    assert torch.equal(net(x), again(x))


def test_dice_loss_on_hand_computed_cases():
    """monai.losses.DiceLoss(sigmoid=True) with its defaults (include_background, smooth_nr = smooth_dr = 1e-5, reduction mean, batch False):
    per (sample, channel) 1 - (2 sum(p y) + 1e-5) / (sum(p) + sum(y) + 1e-5), then the mean -- on 2 x 2 maps worked out by hand."""
    logit = lambda p: math.log(p / (1 - p))
    dice = losses.DiceLoss(sigmoid=True)
    # sample 0: p = 1/2 everywhere, y = [[1, 0], [0, 1]]: intersection 1, sum p 2, sum y 2
    # sample 1: p = [[3/4, 1/4], [1/4, 3/4]], all-zero label: intersection 0, sum p 2, sum y 0
    x = torch.tensor([[[[0.0, 0.0], [0.0, 0.0]]], [[[logit(0.75), logit(0.25)], [logit(0.25), logit(0.75)]]]], dtype=torch.float64)
    y = torch.tensor([[[[1.0, 0.0], [0.0, 1.0]]], [[[0.0, 0.0], [0.0, 0.0]]]], dtype=torch.float64)
    want0 = 1 - (2 * 1.0 + 1e-5) / (2.0 + 2.0 + 1e-5)
    want1 = 1 - (0.0 + 1e-5) / (2.0 + 0.0 + 1e-5)
    assert abs(float(dice(x[:1], y[:1])) - want0) < 1e-12
    assert abs(float(dice(x[1:], y[1:])) - want1) < 1e-12                      # the all-zero-label sample: ~1 - 5e-6, not 0 / 0
    assert abs(float(dice(x, y)) - (want0 + want1) / 2) < 1e-12               # mean over samples, not over the pooled sums
    # perfect prediction of a non-empty label -> 0 (up to the smoothing), of an EMPTY label with p -> 0: (1e-5) / (1e-5) -> 0 as well
    big = torch.where(y[:1] > 0, torch.tensor(40.0, dtype=torch.float64), torch.tensor(-40.0, dtype=torch.float64))
    assert float(dice(big, y[:1])) < 1e-9
    assert float(dice(torch.full_like(x[1:], -800.0), y[1:])) < 1e-12
    # DiceBCELoss (reference utils/losses.py:111-121) = (Dice + BCEWithLogits) / 2
    bce = torch.nn.functional.binary_cross_entropy_with_logits(x, y)
    assert abs(float(losses.DiceBCELoss(True)(x, y)) - (float(dice(x, y)) + float(bce)) / 2) < 1e-12
