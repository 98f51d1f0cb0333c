"""Anti-aliased resampling layers and reflection pad of the GAN networks on the streaming HIP kernels (csrc/blur.hip).

Reference: models/networks.py:244-262 (Upsample), :264-289 (Downsample), nn.ReflectionPad2d of ResnetBlock /
ResnetGenerator (:366-368, :404-421). Every op is one kernel per direction; tensors may be NCHW (``layout="nchw"``) or
NHWC (``layout="nhwc"``), float32 or bfloat16. There is no torch fallback on a GPU tensor: a missing extension raises.
"""
import ctypes

import torch

from .. import _native

_OPS = {
    "reflect_pad": ("octa_reflect_pad_fwd", "octa_reflect_pad_bwd"),
    "blur_down": ("octa_blur_down_fwd", "octa_blur_down_bwd"),
    "blur_up": ("octa_blur_up_fwd", "octa_blur_up_bwd"),
}


def _out_hw(op, h, w, pad):
    if op == "reflect_pad":
        return h + 2 * pad, w + 2 * pad
    if op == "blur_down":
        return (h - 1) // 2 + 1, (w - 1) // 2 + 1
    return 2 * h, 2 * w


def _planes(shape, layout):
    """(B, H, W, C) of the kernels' plane view."""
    if layout == "nchw":
        n, c, h, w = shape
        return n * c, h, w, 1
    n, h, w, c = shape
    return n, h, w, c


def _launch(name, src, dst, b, h, w, c, pad):
    dtype = {torch.float32: 0, torch.bfloat16: 1}[src.dtype]
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    args = [_native.ctx(src.device.index), p(src), p(dst), dtype, b, h, w, c]
    if pad is not None:
        args.append(pad)
    rc = getattr(_native.lib(), name)(*args, _native.current_stream_ptr())
    _native.check(rc, name)


class _Resample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, op, layout, pad):
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        x = x.contiguous()
        b, h, w, c = _planes(x.shape, layout)
        ho, wo = _out_hw(op, h, w, pad)
        shape = (x.shape[0], x.shape[1], ho, wo) if layout == "nchw" else (x.shape[0], ho, wo, x.shape[3])
        y = torch.empty(shape, dtype=x.dtype, device=x.device)
        _launch(_OPS[op][0], x, y, b, h, w, c, pad if op == "reflect_pad" else None)
        ctx.op, ctx.layout, ctx.pad, ctx.in_shape = op, layout, pad, x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        if dy.dtype not in (torch.float32, torch.bfloat16):
            dy = dy.float()
        b, h, w, c = _planes(ctx.in_shape, ctx.layout)
        dx = torch.empty(ctx.in_shape, dtype=dy.dtype, device=dy.device)
        _launch(_OPS[ctx.op][1], dy, dx, b, h, w, c, ctx.pad if ctx.op == "reflect_pad" else None)
        return dx, None, None, None


def reflect_pad(x, pad, layout="nchw"):
    """nn.ReflectionPad2d(pad) of an NCHW or NHWC GPU tensor."""
    return _Resample.apply(x, "reflect_pad", layout, int(pad))


def blur_down(x, layout="nchw"):
    """Downsample(filt_size=3, stride=2, pad_type='reflect') of the reference (networks.py:264-289)."""
    return _Resample.apply(x, "blur_down", layout, 0)


def blur_up(x, layout="nchw"):
    """Upsample(filt_size=4, stride=2, pad_type='repl') of the reference (networks.py:244-262)."""
    return _Resample.apply(x, "blur_up", layout, 0)
