"""torch.autograd bindings of the hand-written HIP kernels used by the training step."""
import ctypes

import torch

from .. import _native


class _InstNormLReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, slope, eps):
        x = x.contiguous()
        B, C = x.shape[0], x.shape[1]
        hw = x.numel() // (B * C)
        dtype = {torch.float32: 0, torch.bfloat16: 1}[x.dtype]
        y = torch.empty_like(x)
        mean = torch.empty(B * C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        w = weight.float().contiguous() if weight is not None else None
        b = bias.float().contiguous() if bias is not None else None
        rc = _native.lib().octa_instnorm_lrelu_fwd(
            _native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(y.data_ptr()),
            ctypes.c_void_p(w.data_ptr()) if w is not None else None, ctypes.c_void_p(b.data_ptr()) if b is not None else None,
            ctypes.c_void_p(mean.data_ptr()), ctypes.c_void_p(rstd.data_ptr()), B, C, hw, dtype, float(slope), float(eps),
            _native.current_stream_ptr())
        _native.check(rc, "octa_instnorm_lrelu_fwd")
        ctx.save_for_backward(x, w, b, mean, rstd)
        ctx.slope, ctx.has_w, ctx.has_b = float(slope), weight is not None, bias is not None
        ctx.w_dtype = weight.dtype if weight is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        B, C = x.shape[0], x.shape[1]
        hw = x.numel() // (B * C)
        dtype = {torch.float32: 0, torch.bfloat16: 1}[x.dtype]
        dx = torch.empty_like(x)
        dw = torch.empty(C, dtype=torch.float32, device=x.device) if ctx.has_w else None
        db = torch.empty(C, dtype=torch.float32, device=x.device) if ctx.has_b else None
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        rc = _native.lib().octa_instnorm_lrelu_bwd(_native.ctx(x.device.index), p(x), p(dy), p(w), p(b), p(mean), p(rstd), p(dx), p(dw), p(db),
                                                   B, C, hw, dtype, ctx.slope, _native.current_stream_ptr())
        _native.check(rc, "octa_instnorm_lrelu_bwd")
        return dx, (dw.to(ctx.w_dtype) if dw is not None else None), (db.to(ctx.w_dtype) if db is not None else None), None, None


def instance_norm_leaky_relu(x, weight, bias, negative_slope=0.01, eps=1e-5):
    """InstanceNorm2d(affine) followed by LeakyReLU in one HIP pass pair (CUDA tensors, float32 or bfloat16)."""
    return _InstNormLReLU.apply(x, weight, bias, negative_slope, eps)
