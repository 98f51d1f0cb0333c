"""Convolutions with ONE channel on one side, NHWC bf16, on csrc/thin_conv.hip: the 7x7 stem / head of ResnetGenerator and the
4x4 stem / head of NLayerDiscriminator (reference models/networks.py:360-368, :433-442, which leaves them to the vendor library).

For y = conv2d(x, w, pad) with stride 1 and t = (ky, kx):
  y[p]      = sum_t x[p + t - pad] . w[t]
  dx[q]     = sum_t dy[q - t + pad] . w[t]          = the convolution of dy with the FLIPPED kernel and pad' = K - 1 - pad
  dw[t]     = sum_p dy[p] (x) x[p + t - pad]
The library has three kernels (include/octa_hip.h): expand (1 -> C), squeeze (C -> 1) and wgrad
(g[c][t] = sum_q a[q][c] s[q + t - pad], a wide, s one channel). With one input channel: forward = expand, dx = squeeze of dy, dw =
wgrad(a = dy, s = x). With one output channel: forward = squeeze, dx = expand of dy, and substituting q = p + t - pad in dw[t][c] =
sum_p dy[p] x[p + t - pad][c] gives sum_q x[q][c] dy[q + t' - pad'] with t' the flipped tap: wgrad(a = x, s = dy, flip, pad')."""
import ctypes

import torch

from .. import _native


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _w2d(weight):
    """Conv2d weight [C, 1, K, K] or [1, C, K, K] -> fp32 [C][K*K] (a view for fp32 parameters)."""
    k = weight.shape[-1]
    return weight.detach().reshape(-1, k * k).float().contiguous()


def _expand(s, w2d, bias, k, pad, flip, slope):
    n, hs, ws = s.shape
    c = w2d.shape[0]
    out = torch.empty((n, hs + 2 * pad - k + 1, ws + 2 * pad - k + 1, c), dtype=torch.bfloat16, device=s.device)
    _native.check(_native.lib().octa_thinconv_expand(_native.ctx(s.device.index), _ptr(s), _ptr(w2d), _ptr(bias), _ptr(out), n, hs, ws, c, k, pad,
                                                     int(flip), float(slope), _native.current_stream_ptr()), "octa_thinconv_expand")
    return out


def _squeeze(a, w2d, bias, k, pad, flip):
    n, ha, wa, c = a.shape
    out = torch.empty((n, ha + 2 * pad - k + 1, wa + 2 * pad - k + 1), dtype=torch.bfloat16, device=a.device)
    _native.check(_native.lib().octa_thinconv_squeeze(_native.ctx(a.device.index), _ptr(a), _ptr(w2d), _ptr(bias), _ptr(out), n, ha, wa, c, k, pad,
                                                      int(flip), _native.current_stream_ptr()), "octa_thinconv_squeeze")
    return out


def _wgrad(a, s, k, pad, flip, want_asum):
    n, ha, wa, c = a.shape
    lib = _native.lib()
    scratch = torch.empty(int(lib.octa_thinconv_wgrad_scratch_floats(n, ha, c, k)), dtype=torch.float32, device=a.device)
    g = torch.empty((c, k * k), dtype=torch.float32, device=a.device)
    asum = torch.empty(c, dtype=torch.float32, device=a.device) if want_asum else None
    _native.check(lib.octa_thinconv_wgrad(_native.ctx(a.device.index), _ptr(a), _ptr(s), _ptr(scratch), _ptr(g), _ptr(asum), n, ha, wa, s.shape[1],
                                          s.shape[2], c, k, pad, int(flip), _native.current_stream_ptr()), "octa_thinconv_wgrad")
    return g, asum


def _bf16c(t):
    return t.to(torch.bfloat16).contiguous()


class _ConvFrom1(torch.autograd.Function):
    """x [N, H, W] (one channel) -> y [N, Ho, Wo, C]; optional LeakyReLU(slope) fused behind the bias."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad, slope):
        k = weight.shape[-1]
        xs = _bf16c(x)
        y = _expand(xs, _w2d(weight), None if bias is None else bias.detach().float().contiguous(), k, pad, False, slope)
        ctx.save_for_backward(xs, weight, y if slope != 1.0 else None)
        ctx.cfg = (k, pad, slope, bias is not None, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, weight, y = ctx.saved_tensors
        k, pad, slope, has_bias, x_dtype = ctx.cfg
        dy = _bf16c(dy)
        if slope != 1.0:                                   # LeakyReLU': the sign of the output is the sign of its input (slope > 0)
            g = torch.empty_like(dy)                       # one launch instead of torch's compare, scale and select (same values: y > 0 ? dy : bf16(dy * slope))
            _native.check(_native.lib().octa_lrelu_bwd_bf16(_native.ctx(dy.device.index), _ptr(y), _ptr(dy), _ptr(g), dy.numel(), float(slope),
                                                           _native.current_stream_ptr()), "octa_lrelu_bwd_bf16")
            dy = g
        dx = _squeeze(dy, _w2d(weight), None, k, k - 1 - pad, True).to(x_dtype) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1] or (has_bias and ctx.needs_input_grad[2]):
            g, asum = _wgrad(dy, xs, k, pad, False, has_bias)
            dw = g.view(weight.shape).to(weight.dtype)
            db = asum if has_bias else None
        return dx, dw, db, None, None


class _ConvTo1(torch.autograd.Function):
    """x [N, H, W, C] -> y [N, Ho, Wo] (one channel)."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad):
        k = weight.shape[-1]
        xs = _bf16c(x)
        y = _squeeze(xs, _w2d(weight), None if bias is None else bias.detach().float().contiguous(), k, pad, False)
        ctx.save_for_backward(xs, weight)
        ctx.cfg = (k, pad, bias is not None, x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, weight = ctx.saved_tensors
        k, pad, has_bias, x_dtype = ctx.cfg
        dy = _bf16c(dy)
        dx = _expand(dy, _w2d(weight), None, k, k - 1 - pad, True, 1.0).to(x_dtype) if ctx.needs_input_grad[0] else None
        dw = db = None
        if ctx.needs_input_grad[1]:
            g, _ = _wgrad(xs, dy, k, k - 1 - pad, True, False)
            dw = g.view(weight.shape).to(weight.dtype)
        if has_bias and ctx.needs_input_grad[2]:
            db = dy.float().sum().reshape(1)
        return dx, dw, db, None


def supported(conv):
    """nn.Conv2d with one channel on one side that the thin kernels cover."""
    return (conv.kernel_size in ((4, 4), (7, 7)) and conv.stride == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.padding[0] == conv.padding[1] and conv.padding[0] < conv.kernel_size[0] and isinstance(conv.padding, tuple)
            and ((conv.in_channels == 1 and conv.out_channels % 64 == 0 and conv.out_channels <= 1024 and 256 % (conv.out_channels // 8) == 0)
                 or (conv.out_channels == 1 and conv.in_channels % 64 == 0 and conv.in_channels <= 1024 and 256 % (conv.in_channels // 8) == 0))
            and conv.kernel_size[0] ** 2 * max(conv.in_channels, conv.out_channels) * 4 <= 60 * 1024)


def conv_from_1(x_nhw, conv, slope=1.0):
    """Conv2d(1 -> C): x [N, H, W] -> bf16 [N, Ho, Wo, C] (+ LeakyReLU(slope) when slope != 1)."""
    return _ConvFrom1.apply(x_nhw, conv.weight, conv.bias, conv.padding[0], float(slope))


def conv_to_1(x_nhwc, conv):
    """Conv2d(C -> 1): x [N, H, W, C] -> bf16 [N, Ho, Wo]."""
    return _ConvTo1.apply(x_nhwc, conv.weight, conv.bias, conv.padding[0])
