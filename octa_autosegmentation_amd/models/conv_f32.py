"""Single-precision convolutions on the matrix cores (csrc/conv_f32.hip, `octa_conv2d_f32_nchw`) for the reference's paths that run
WITHOUT mixed precision: test.py:79 / validate.py evaluate `model.inference` outside autocast, so DynUNet's torch.nn.Conv2d /
ConvTranspose2d layers (models/networks.py:6 -> MONAI) compute in fp32 there. Forward only: a pass that needs gradients in fp32
(`General.amp: false` training) stays on the torch modules.

Weights are re-laid-out once per weight version as [Cin][K*K][Cout] (output channel innermost: the kernel's weight slice loads are
then contiguous); a 2x2 stride-2 transposed convolution is one launch on the packed tensor [Cin][4][Cout]
(`octa_convtranspose2x2_f32_nchw`)."""
import ctypes

import torch

from .. import _native

_EPOCH = [0]         # bumped by invalidate_packs(): every packed copy made before is stale


def invalidate_packs():
    """Weights were rewritten through .data / load_state_dict (no version bump): forget every packed copy."""
    _EPOCH[0] += 1


def _packed(weight, transposed):
    """The kernel's weight layout, cached ON the parameter object (epoch, version, storage address, orientation): a cache keyed by
    id(weight) served a stale copy when a new parameter was given the id and the storage address of a collected one (two DynUNets
    built one after the other in a test run: logits off by whole units, about one full test run in eight)."""
    hit = getattr(weight, "_octa_f32_pack", None)
    if hit is not None and hit[0] == _EPOCH[0] and hit[1] == weight._version and hit[2] == weight.data_ptr() and hit[3] == transposed:
        return hit[4]
    w = weight.detach().float()
    if transposed:          # [Cin][Cout][k][k] -> [Cin][k*k][Cout]
        p = w.permute(0, 2, 3, 1).reshape(w.shape[0], w.shape[2] * w.shape[3], w.shape[1]).contiguous()
    else:                   # [Cout][Cin][K][K] -> [Cin][K*K][Cout]
        p = w.permute(1, 2, 3, 0).reshape(w.shape[1], w.shape[2] * w.shape[3], w.shape[0]).contiguous()
    weight._octa_f32_pack = (_EPOCH[0], weight._version, weight.data_ptr(), transposed, p)
    return p


def _launch(x, wp, wp_offset, bias, y, Cout, cout_w, K, stride, pad, Ho, Wo, osc=1, ooy=0, oox=0):
    N, Cin, H, W = x.shape
    rc = _native.lib().octa_conv2d_f32_nchw(
        _native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wp.data_ptr() + 4 * wp_offset),
        ctypes.c_void_p(bias.data_ptr()) if bias is not None else None, ctypes.c_void_p(y.data_ptr()),
        N, Cin, H, W, Cout, cout_w, K, stride, pad, Ho, Wo, osc, ooy, oox, _native.current_stream_ptr())
    _native.check(rc, "octa_conv2d_f32_nchw")


def supported(conv):
    """torch.nn.Conv2d / ConvTranspose2d layers this path evaluates (everything DynUNet-S holds)."""
    if conv.groups != 1 or tuple(conv.dilation) != (1, 1):
        return False
    k, s, p = tuple(conv.kernel_size), tuple(conv.stride), tuple(conv.padding)
    if k[0] != k[1] or s[0] != s[1] or p[0] != p[1]:
        return False
    if isinstance(conv, torch.nn.ConvTranspose2d):
        return k == s and k[0] in (1, 2) and p == (0, 0) and tuple(conv.output_padding) == (0, 0) and conv.bias is None
    if isinstance(conv, torch.nn.Conv2d):
        # zero padding up to half the kernel: DynUNet's "same" layers, the GAN networks' 3x3 / 7x7 layers behind a reflection pad (padding 0)
        # and PatchGAN's 4x4 layers with padding 1
        return (k[0], s[0]) in ((1, 1), (3, 1), (3, 2), (4, 1), (7, 1)) and 0 <= p[0] <= k[0] // 2 and conv.padding_mode == "zeros"
    return False


def applies(conv, x):
    """The fp32 matrix-core path takes a call when it is plain fp32 on the GPU and no gradient is recorded."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and supported(conv)):
        return False
    if torch.is_autocast_enabled():
        return False
    return not (torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad))


def forward(conv, x):
    """conv(x) for a supported layer: fp32 in, fp32 out, NCHW."""
    x = x.contiguous()
    N, Cin, H, W = x.shape
    if isinstance(conv, torch.nn.ConvTranspose2d):
        k, Cout = conv.kernel_size[0], conv.out_channels
        wp = _packed(conv.weight, True)
        y = torch.empty((N, Cout, H * k, W * k), dtype=torch.float32, device=x.device)
        if k == 2:          # one launch, both output parities of a row pair written as 8-byte pairs
            rc = _native.lib().octa_convtranspose2x2_f32_nchw(
                _native.ctx(x.device.index), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(y.data_ptr()),
                N, Cin, H, W, Cout, _native.current_stream_ptr())
            _native.check(rc, "octa_convtranspose2x2_f32_nchw")
            return y
        for a in range(k):
            for b in range(k):
                _launch(x, wp, (a * k + b) * Cout, None, y, Cout, k * k * Cout, 1, 1, 0, H, W, k, a, b)
        return y
    K, s, Cout = conv.kernel_size[0], conv.stride[0], conv.out_channels
    pad = conv.padding[0]
    Ho, Wo = (H + 2 * pad - K) // s + 1, (W + 2 * pad - K) // s + 1
    wp = _packed(conv.weight, False)
    bias = conv.bias.detach().float().contiguous() if conv.bias is not None else None
    y = torch.empty((N, Cout, Ho, Wo), dtype=torch.float32, device=x.device)
    _launch(x, wp, 0, bias, y, Cout, Cout, K, s, pad, Ho, Wo)
    return y
