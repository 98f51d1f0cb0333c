"""Host side of the hand-written MFMA convolution (csrc/conv.hip): NHWC bf16 tensors in, NHWC bf16 out.

Layouts: activations [N, H, W, C] contiguous bfloat16; weights "tap-major" [9, Cout, Cin] bfloat16 (tap = 3*r+s).
`pack_weight` / `pack_weight_dgrad` convert a torch conv weight [Cout, Cin, 3, 3] (models/networks.py, MONAI
DynUNet naming) into the forward layout and into the layout whose forward pass IS the data gradient.
"""
import ctypes

import torch

from .. import _native


def pack_weight(w):
    """[Cout, Cin, 3, 3] -> [9, Cout, Cin] bf16."""
    co, ci = w.shape[0], w.shape[1]
    return w.permute(2, 3, 0, 1).reshape(9, co, ci).to(torch.bfloat16).contiguous()


def pack_weight_dgrad(w):
    """[Cout, Cin, 3, 3] -> [9, Cin, Cout] bf16 with the taps flipped: conv3x3(dy, this) = dL/dx."""
    co, ci = w.shape[0], w.shape[1]
    return w.flip(2, 3).permute(2, 3, 1, 0).reshape(9, ci, co).to(torch.bfloat16).contiguous()


def conv3x3_nhwc(x, wt, stride=1, in_dilation=1):
    """x [N,H,W,Cin] bf16, wt [9,Cout,Cin] bf16 -> [N,Ho,Wo,Cout] bf16 (padding 1)."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and x.dim() == 4
    assert wt.dtype == torch.bfloat16 and wt.is_contiguous() and wt.shape[0] == 9 and wt.shape[2] == x.shape[3]
    n, h, w, cin = x.shape
    cout = wt.shape[1]
    ho = (h * in_dilation - 1) // stride + 1
    wo = (w * in_dilation - 1) // stride + 1
    y = torch.empty((n, ho, wo, cout), dtype=torch.bfloat16, device=x.device)
    rc = _native.lib().octa_conv3x3_nhwc_fwd(_native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()),
                                             ctypes.c_void_p(wt.data_ptr()), ctypes.c_void_p(y.data_ptr()), n, h, w, cin, cout,
                                             int(stride), int(in_dilation), _native.current_stream_ptr())
    _native.check(rc, "octa_conv3x3_nhwc_fwd")
    return y


def conv3x3_nhwc_wgrad(x, dy):
    """x [N,H,W,Cin] bf16, dy [N,H,W,Cout] bf16 (stride-1 layer) -> dW as a torch conv weight gradient
    [Cout, Cin, 3, 3] float32."""
    assert x.is_cuda and x.dtype == torch.bfloat16 and x.is_contiguous() and dy.dtype == torch.bfloat16 and dy.is_contiguous()
    n, h, w, cin = x.shape
    assert dy.shape[:3] == x.shape[:3]
    cout = dy.shape[3]
    dw = torch.empty((9, cout, cin), dtype=torch.float32, device=x.device)
    rc = _native.lib().octa_conv3x3_nhwc_wgrad(_native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(dy.data_ptr()),
                                               ctypes.c_void_p(dw.data_ptr()), n, h, w, cin, cout, _native.current_stream_ptr())
    _native.check(rc, "octa_conv3x3_nhwc_wgrad")
    return dw.view(3, 3, cout, cin).permute(2, 3, 0, 1)
