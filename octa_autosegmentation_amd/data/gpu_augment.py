"""The training configs' augmentation chain on the GPU, for a whole batch that never leaves HBM.

Reference: `Train.data_augmentation` of configs/config_ves_seg-S.yml:28-102, instantiated by
data/data_transforms.py:587-611 (MONAI dictionary transforms on the CPU, one sample at a time in loader workers):
  LoadGraphAndFilterByRandomRadiusd -> ScaleIntensityd(0,1) -> EnsureChannelFirstd -> Resized(1216,1216, bilinear)
  -> RandFlipd(0.5, axes [0,1]) -> RandRotate90d(0.75) -> RandRotated(prob 1, +-range_x, zeros) -> AsDiscreted(label, 0.1)
  -> CastToTyped.
Here the rasteriser's uint8 batches (pipeline.TripleGenerator) go through two HIP kernels (csrc/augment.hip); the random
decisions are drawn on the host from one numpy RandomState in the order MONAI's `randomize` methods draw them
(flip: rand() < p; rot90: rand() < p, then randint(3) + 1; rotate: rand() < p, then uniform(-r, r)), image and label of a
sample share them. MONAI is not installed here: its per-transform random streams are not reproduced (parity unpinned),
the geometry is pinned against the torch ops MONAI delegates to (tests/test_augment_gpu.py).
"""
import ctypes

import numpy as np
import torch

from .. import _native


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def resize_bilinear(x, size, mul=None, add=None):
    """x: CUDA uint8 / float32 [B,h,w] -> float32 [B,H,W] (torch bilinear, align_corners=False); optional per-image
    affine map of the source values (ScaleIntensity)."""
    assert x.is_cuda and x.dim() == 3 and x.is_contiguous() and x.dtype in (torch.uint8, torch.float32)
    B, h, w = x.shape
    out = torch.empty((B, int(size[0]), int(size[1])), dtype=torch.float32, device=x.device)
    rc = _native.lib().octa_resize_bilinear(_native.ctx(x.device.index), _p(x), 0 if x.dtype == torch.uint8 else 1, B, h, w, _p(out),
                                            out.shape[1], out.shape[2], _p(mul), _p(add), _native.current_stream_ptr())
    _native.check(rc, "octa_resize_bilinear")
    return out


def resize_bilinear_bwd(dy, size):
    """Adjoint of resize_bilinear: dy CUDA float32 [B,H,W] -> float32 [B,h,w] with size = (h, w)."""
    assert dy.is_cuda and dy.dim() == 3 and dy.is_contiguous() and dy.dtype == torch.float32
    B, H, W = dy.shape
    dx = torch.empty((B, int(size[0]), int(size[1])), dtype=torch.float32, device=dy.device)
    rc = _native.lib().octa_resize_bilinear_bwd(_native.ctx(dy.device.index), _p(dy), B, dx.shape[1], dx.shape[2], H, W, _p(dx), _native.current_stream_ptr())
    _native.check(rc, "octa_resize_bilinear_bwd")
    return dx


class BilinearResize(torch.autograd.Function):
    """F.interpolate(x, size, mode="bilinear") (align_corners False) on [B, C, h, w] CUDA tensors through csrc/augment.hip, both ways."""

    @staticmethod
    def forward(ctx, x, size):
        B, C, h, w = x.shape
        ctx.in_shape, ctx.in_dtype = (h, w), x.dtype
        y = resize_bilinear(x.reshape(B * C, h, w).float().contiguous(), size)
        return y.view(B, C, int(size[0]), int(size[1])).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        B, C, H, W = dy.shape
        dx = resize_bilinear_bwd(dy.reshape(B * C, H, W).float().contiguous(), ctx.in_shape)
        return dx.view(B, C, *ctx.in_shape).to(ctx.in_dtype), None


def flip_rot90_rotate(x, angle, rot_k=None, flip=None, threshold=None):
    """x: CUDA float32 [B,N,N]; angle float32 [B] (radians), rot_k / flip int32 [B] or None."""
    assert x.is_cuda and x.dim() == 3 and x.shape[1] == x.shape[2] and x.dtype == torch.float32 and x.is_contiguous()
    out = torch.empty_like(x)
    rc = _native.lib().octa_flip_rot90_rotate(_native.ctx(x.device.index), _p(x), _p(out), x.shape[0], x.shape[1], _p(angle), _p(rot_k), _p(flip),
                                              float(threshold if threshold is not None else 0.0), 0 if threshold is None else 1,
                                              _native.current_stream_ptr())
    _native.check(rc, "octa_flip_rot90_rotate")
    return out


def background_noise(img, noise, u, out_dtype=torch.float64):
    """AddRandomBackgroundNoised on device tensors of equal shape: img / noise float32, u float64 (numpy's uniform factors).
    Returns max(img, noise * u) as float64 (the reference's promoted dtype) or float32."""
    assert img.is_cuda and img.shape == noise.shape == u.shape and u.dtype == torch.float64
    img, noise, u = img.contiguous().float(), noise.contiguous().float(), u.contiguous()
    out = torch.empty(img.shape, dtype=out_dtype, device=img.device)
    o64, o32 = (_p(out), None) if out_dtype == torch.float64 else (None, _p(out))
    rc = _native.lib().octa_background_noise(_native.ctx(img.device.index), _p(img), _p(noise), _p(u), img.numel(), o64, o32, _native.current_stream_ptr())
    _native.check(rc, "octa_background_noise")
    return out


def speckle_brightness(img, grid9, u):
    """SpeckleBrightnesd on a device batch: img float32 [B,H,W], grid9 float32 [B,9,9] (control values in [0.5, 1)), u float32 [B,H,W]."""
    assert img.is_cuda and img.dim() == 3 and grid9.shape == (img.shape[0], 9, 9) and u.shape == img.shape
    img, grid9, u = img.contiguous().float(), grid9.contiguous().float(), u.contiguous().float()
    out = torch.empty_like(img)
    mm = torch.empty((img.shape[0], 2), dtype=torch.int32, device=img.device)
    rc = _native.lib().octa_speckle_brightness(_native.ctx(img.device.index), _p(img), _p(grid9), _p(u), img.shape[0], img.shape[1], img.shape[2],
                                               _p(out), _p(mm), _native.current_stream_ptr())
    _native.check(rc, "octa_speckle_brightness")
    return out


class GpuSegAugmentation:
    """Batched replacement of the `data_augmentation` list of a segmentation config (the entries after the graph loader)."""

    def __init__(self, aug_config, seed=None):
        self.size, self.flip_p, self.rot90_p, self.rot_p, self.rot_range, self.threshold = None, 0.0, 0.0, 0.0, 0.0, None
        self.scale = None
        for d in aug_config:
            name = d["name"]
            if name in ("LoadGraphAndFilterByRandomRadiusd", "EnsureChannelFirstd", "CastToTyped"):
                continue
            if name == "ScaleIntensityd":
                self.scale = (float(d.get("minv", 0.0)), float(d.get("maxv", 1.0)))
            elif name == "Resized":
                if d.get("mode", "bilinear") != "bilinear":
                    raise NotImplementedError("Resized: only mode bilinear is on the GPU path")
                self.size = [int(v) for v in d["spatial_size"]]
            elif name == "RandFlipd":
                if sorted(d.get("spatial_axis", [0, 1])) != [0, 1]:
                    raise NotImplementedError("RandFlipd: the configs flip both axes")
                self.flip_p = float(d.get("prob", 0.1))
            elif name == "RandRotate90d":
                self.rot90_p = float(d.get("prob", 0.1))
            elif name == "RandRotated":
                if d.get("padding_mode", "border") != "zeros":
                    raise NotImplementedError("RandRotated: only padding_mode zeros is on the GPU path")
                self.rot_p, self.rot_range = float(d.get("prob", 0.1)), float(d.get("range_x", 0.0))
            elif name == "AsDiscreted":
                self.threshold = float(d["threshold"])
            else:
                raise NotImplementedError(f"transform {name} is not part of the GPU augmentation chain")
        # one generator PER random transform, all seeded alike -- what get_data_augmentations does with MONAI's Randomizable objects
        # (data_transforms.py:606-607) and what data/data_transforms.py's RandFlipd / RandRotate90d / RandRotated do: the fused chain
        # and the generic per-sample transforms take the same decisions for the same seed (tests/test_training_cli_gpu.py)
        self.R_flip, self.R_rot90, self.R_rot = (np.random.RandomState(seed) for _ in range(3))

    def draw(self, batch):
        """Per-sample (flip, k, angle), every transform drawing from its own stream in the order its `randomize` draws."""
        flip = np.zeros(batch, np.int32)
        k = np.zeros(batch, np.int32)
        ang = np.zeros(batch, np.float32)
        for b in range(batch):
            flip[b] = self.R_flip.rand() < self.flip_p
            kk = self.R_rot90.randint(3) + 1
            if self.R_rot90.rand() < self.rot90_p:
                k[b] = kk
            if self.R_rot.rand() < self.rot_p:
                ang[b] = self.R_rot.uniform(low=-self.rot_range, high=self.rot_range)
                self.R_rot.uniform(low=0.0, high=0.0)
                self.R_rot.uniform(low=0.0, high=0.0)
        return flip, k, ang

    def _scale_map(self, x):
        if self.scale is None:
            return None, None
        # per-sample extrema in two passes (rows first): a reduction of [B, h*w] to B values runs on B workgroups -- 0.37 ms per call for
        # four 1216^2 labels, 1.5 ms of reductions per training step on the loader's stream -- and the uint8 label needs no fp32 copy
        rows = x.reshape(x.shape[0], -1, x.shape[-1]) if x.dim() >= 3 else x.reshape(x.shape[0], 1, -1)
        mn, mx = rows.amin(dim=2).amin(dim=1).float(), rows.amax(dim=2).amax(dim=1).float()
        lo, hi = self.scale
        span = mx - mn
        mul = torch.where(span > 0, (hi - lo) / span, torch.zeros_like(span))     # MONAI: a constant image maps to minv
        return mul.contiguous(), (lo - mn * mul).contiguous()

    def __call__(self, image, label):
        """image, label: CUDA uint8 / float32 [B,h,w] (rasteriser output). -> dict(image, label) float32 [B,1,H,W]."""
        B = image.shape[0]
        flip, k, ang = self.draw(B)
        dev = image.device
        flip_t, k_t, ang_t = (torch.from_numpy(a).to(dev) for a in (flip, k, ang))
        out = {}
        for key, x, thr in (("image", image, None), ("label", label, self.threshold)):
            mul, add = self._scale_map(x)
            size = self.size or list(x.shape[1:])
            y = resize_bilinear(x.contiguous(), size, mul, add)
            out[key] = flip_rot90_rotate(y, ang_t, k_t, flip_t, thr).unsqueeze(1)
        out["params"] = dict(flip=flip, rot_k=k, angle=ang)
        return out
