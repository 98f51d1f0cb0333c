"""GPU parity of the augmentation kernels (csrc/augment.hip) against the torch ops MONAI's transforms delegate to:
F.interpolate(bilinear), torch.flip / torch.rot90, F.affine_grid + F.grid_sample. fp32, tolerance 1e-5 (same formulas,
fused multiply-adds may differ in the last bits)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_resize_matches_torch_interpolate(hip_lib_built):
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.data import gpu_augment
    g = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randint(0, 256, (3, 76, 76), device="cuda", generator=g, dtype=torch.uint8)
    got = gpu_augment.resize_bilinear(x, [304, 304])
    want = F.interpolate(x.float().unsqueeze(1), size=(304, 304), mode="bilinear", align_corners=False).squeeze(1)
    assert (got - want).abs().max().item() <= 1e-3          # values up to 255
    xf = torch.rand(2, 50, 70, device="cuda", generator=g)
    mul, add = torch.tensor([2.0, 0.5], device="cuda"), torch.tensor([-1.0, 3.0], device="cuda")
    got = gpu_augment.resize_bilinear(xf, [125, 91], mul, add)
    want = F.interpolate((xf * mul[:, None, None] + add[:, None, None]).unsqueeze(1), size=(125, 91), mode="bilinear", align_corners=False).squeeze(1)
    assert (got - want).abs().max().item() <= 1e-5
    same = gpu_augment.resize_bilinear(xf, [50, 70])
    assert (same - xf).abs().max().item() <= 1e-6           # Resized to the same size is the identity


def test_resize_adjoint_matches_torch_autograd(hip_lib_built):
    """octa_resize_bilinear_bwd (round 4): the gradient of F.interpolate(bilinear) for the GAN-seg model's 304 -> 1216 up-sampling, a
    non-integer ratio, a down-sampling and the identity; through the autograd function on [B, C, h, w] in fp32 and bf16."""
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.data import gpu_augment
    g = torch.Generator(device="cuda").manual_seed(4)
    for (h, w, H, W) in [(304, 304, 1216, 1216), (50, 70, 125, 91), (96, 64, 48, 40), (33, 17, 33, 17), (5, 3, 64, 64)]:
        x = torch.rand(2, 1, h, w, device="cuda", generator=g, requires_grad=True)
        dy = torch.randn(2, 1, H, W, device="cuda", generator=g)
        F.interpolate(x, size=(H, W), mode="bilinear", align_corners=False).backward(dy)
        want = x.grad.clone()
        got = gpu_augment.resize_bilinear_bwd(dy.reshape(2, H, W).contiguous(), (h, w)).view(2, 1, h, w)
        assert (got - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item()), (h, w, H, W)
        x2 = x.detach().clone().requires_grad_(True)
        y2 = gpu_augment.BilinearResize.apply(x2, (H, W))
        assert (y2 - F.interpolate(x.detach(), size=(H, W), mode="bilinear", align_corners=False)).abs().max().item() <= 1e-5
        y2.backward(dy)
        assert (x2.grad - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    xb = torch.rand(3, 2, 40, 40, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True)
    yb = gpu_augment.BilinearResize.apply(xb, (160, 160))
    assert yb.dtype == torch.bfloat16 and yb.shape == (3, 2, 160, 160)
    yb.float().sum().backward()
    assert xb.grad.dtype == torch.bfloat16 and abs(float(xb.grad.float().sum()) - 3 * 2 * 160 * 160) < 0.02 * 3 * 2 * 160 * 160      # the weights of every output pixel sum to one


def test_flip_rot90_rotate_matches_torch(hip_lib_built):
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.data import gpu_augment
    g = torch.Generator(device="cuda").manual_seed(2)
    B, N = 8, 96
    x = torch.rand(B, N, N, device="cuda", generator=g)
    flip = torch.tensor([0, 1, 0, 1, 0, 1, 0, 1], dtype=torch.int32, device="cuda")
    k = torch.tensor([0, 0, 1, 1, 2, 2, 3, 3], dtype=torch.int32, device="cuda")
    ang = torch.tensor([0.0, 0.1, -0.17, 0.05, 0.17, -0.03, 0.12, -0.1], device="cuda")
    got = gpu_augment.flip_rot90_rotate(x, ang, k, flip)
    for b in range(B):
        y = x[b]
        if int(flip[b]):
            y = torch.flip(y, (0, 1))
        y = torch.rot90(y, int(k[b]), (0, 1))
        c, s = float(torch.cos(ang[b])), float(torch.sin(ang[b]))
        theta = torch.tensor([[[c, -s, 0.0], [s, c, 0.0]]], device="cuda")
        grid = F.affine_grid(theta, (1, 1, N, N), align_corners=False)
        want = F.grid_sample(y[None, None], grid, mode="bilinear", padding_mode="zeros", align_corners=False)[0, 0]
        assert (got[b] - want).abs().max().item() <= 2e-5, b
    thr = gpu_augment.flip_rot90_rotate(x, ang, k, flip, threshold=0.5)
    assert set(thr.unique().tolist()) <= {0.0, 1.0}
    assert ((thr > 0.5) != (got >= 0.5)).float().mean().item() < 1e-4


def test_augmentation_chain_from_config(hip_lib_built):
    """The reference's config list drives the chain; image and label of a sample get the same geometry."""
    import torch
    from octa_autosegmentation_amd.data import gpu_augment
    cfg = [{"name": "LoadGraphAndFilterByRandomRadiusd", "keys": ["image", "label"]},
           {"name": "ScaleIntensityd", "keys": ["image", "label"], "minv": 0, "maxv": 1},
           {"name": "EnsureChannelFirstd", "keys": ["image", "label"]},
           {"name": "Resized", "keys": ["image", "label"], "spatial_size": [128, 128], "mode": "bilinear"},
           {"name": "RandFlipd", "keys": ["image", "label"], "prob": 0.5, "spatial_axis": [0, 1]},
           {"name": "RandRotate90d", "keys": ["image", "label"], "prob": 0.75},
           {"name": "RandRotated", "keys": ["image", "label"], "prob": 1, "range_x": 0.1745, "padding_mode": "zeros"},
           {"name": "AsDiscreted", "keys": ["label"], "threshold": 0.1},
           {"name": "CastToTyped", "keys": ["image", "label"], "dtype": "dtype"}]
    aug = gpu_augment.GpuSegAugmentation(cfg, seed=3)
    g = torch.Generator(device="cuda").manual_seed(4)
    img = torch.randint(0, 256, (6, 32, 32), device="cuda", generator=g, dtype=torch.uint8)
    lab = (torch.rand(6, 128, 128, device="cuda", generator=g) > 0.7).to(torch.uint8) * 255
    out = aug(img, lab)
    assert out["image"].shape == (6, 1, 128, 128) and out["label"].shape == (6, 1, 128, 128)
    assert out["image"].dtype == torch.float32 and 0.0 <= float(out["image"].min()) and float(out["image"].max()) <= 1.0 + 1e-6
    assert set(out["label"].unique().tolist()) <= {0.0, 1.0}
    p = out["params"]
    assert p["rot_k"].max() <= 3 and np.abs(p["angle"]).max() <= 0.1745 + 1e-6
    # same seed -> same decisions
    again = gpu_augment.GpuSegAugmentation(cfg, seed=3)(img, lab)
    assert torch.equal(again["image"], out["image"]) and torch.equal(again["label"], out["label"])


def test_noise_kernels_match_the_reference_fixture(hip_lib_built):
    """a17 on the device: the HIP kernels behind SpeckleBrightnesd / AddRandomBackgroundNoised (csrc/augment.hip) against the outputs of
    the reference's own classes (tests/golden/noise_golden.npz, tools/make_golden_noise.py) with the same generator seeds. The
    background blend is exact (one float64 product, one comparison); the speckle map differs from torch's CPU arithmetic by fused
    multiply-adds only (1e-6 of values in [0, 1])."""
    import os
    import torch
    from octa_autosegmentation_amd.data import data_transforms as T
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "noise_golden.npz"))
    for k in range(2):
        img = torch.from_numpy(g[f"speckle_{k}_in"]).cuda()
        torch.manual_seed(100 + k)
        out = T.SpeckleBrightnesd(["image"])({"image": img.clone()})["image"]
        assert out.is_cuda and out.dtype == torch.float32
        assert (out.cpu() - torch.from_numpy(g[f"speckle_{k}_out"])).abs().max().item() <= 1e-6
        np.random.seed(200 + k)
        d = T.AddRandomBackgroundNoised(["image"])({"image": img.clone(), "background": torch.from_numpy(g[f"bg_{k}_noise"]).cuda()})
        assert d["image"].is_cuda and d["image"].dtype == torch.float64 and torch.equal(d["image"].cpu(), torch.from_numpy(g[f"bg_{k}_out"]))
    # batched entry points
    from octa_autosegmentation_amd.data import gpu_augment
    gen = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(3, 64, 80, device="cuda", generator=gen)
    c9 = torch.rand(3, 9, 9, device="cuda", generator=gen) * 0.5 + 0.5
    u = torch.rand(3, 64, 80, device="cuda", generator=gen)
    got = gpu_augment.speckle_brightness(x, c9, u)
    C = torch.nn.functional.interpolate(c9[:, None], size=(64, 80), mode="bilinear")[:, 0]
    v = x * (C - u * (1 - C))
    want = v / v.amax(dim=(1, 2), keepdim=True)
    want = want - want.amin(dim=(1, 2), keepdim=True)
    assert (got - want).abs().max().item() <= 2e-6
    u64 = torch.rand(3, 64, 80, device="cuda", generator=gen, dtype=torch.float64)
    assert torch.equal(gpu_augment.background_noise(x, u, u64), torch.maximum(x.double(), u.double() * u64))
    assert torch.equal(gpu_augment.background_noise(x, u, u64, torch.float32), torch.maximum(x.double(), u.double() * u64).float())
