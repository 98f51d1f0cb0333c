"""GPU: the one-channel-side convolutions of the GAN networks (csrc/thin_conv.hip through models/thin_conv.py) against torch's
conv2d in fp32 on the same bf16-rounded operands: forward, data gradient, weight and bias gradients; tolerance = one bf16 rounding
of the result (outputs / dx) and fp32 summation order (dw, db)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(x_nchw, conv, slope, dy_nchw):
    x = x_nchw.float().requires_grad_(True)
    w = conv.weight.detach().float().requires_grad_(True)
    b = conv.bias.detach().float().requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=False):
        y = torch.nn.functional.conv2d(x, w, b, padding=conv.padding)
    if slope != 1.0:
        y = torch.nn.functional.leaky_relu(y, slope)
    y.backward(dy_nchw.float())
    return y.detach(), x.grad, w.grad, b.grad


def _close(got, want, rtol, what):
    err = (got.float() - want).abs().max().item()
    scale = want.abs().max().item()
    assert err <= rtol * scale + 1e-6, f"{what}: max error {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("k,pad,c,h,w,slope", [(7, 0, 64, 70, 86, 1.0), (4, 1, 64, 61, 75, 0.2), (7, 0, 64, 310, 310, 1.0)])
def test_one_input_channel(hip_lib_built, k, pad, c, h, w, slope):
    """Generator stem (7x7 on the reflect-padded image) and discriminator stem (4x4, padding 1, LeakyReLU(0.2) fused)."""
    from octa_autosegmentation_amd.models import thin_conv
    torch.manual_seed(k * 100 + h)
    conv = torch.nn.Conv2d(1, c, k, 1, pad).cuda()
    assert thin_conv.supported(conv)
    x = torch.randn(3, 1, h, w, device="cuda").to(torch.bfloat16)
    xs = x[:, 0].clone().requires_grad_(True)
    y = thin_conv.conv_from_1(xs, conv, slope)
    dy = torch.randn(y.shape, device="cuda").to(torch.bfloat16)
    y.backward(dy)
    y_ref, dx_ref, dw_ref, db_ref = _ref(x, conv, slope, dy.permute(0, 3, 1, 2))
    _close(y.permute(0, 3, 1, 2), y_ref, 6e-3, "forward")
    _close(xs.grad[:, None], dx_ref, 6e-3, "dx")
    _close(conv.weight.grad, dw_ref, 2e-3 if slope == 1.0 else 1e-2, "dw")      # with the fused LeakyReLU dy * slope is rounded to bf16 once more
    _close(conv.bias.grad, db_ref, 2e-3 if slope == 1.0 else 1e-2, "db")


@pytest.mark.parametrize("k,pad,c,h,w", [(7, 0, 64, 70, 86), (4, 1, 512, 38, 38), (4, 1, 128, 21, 45), (7, 0, 64, 310, 310)])
def test_one_output_channel(hip_lib_built, k, pad, c, h, w):
    """Generator head (7x7, 64 -> 1) and discriminator head (4x4, 512 -> 1)."""
    from octa_autosegmentation_amd.models import thin_conv
    torch.manual_seed(k * 100 + h + c)
    conv = torch.nn.Conv2d(c, 1, k, 1, pad).cuda()
    assert thin_conv.supported(conv)
    x = torch.randn(2, c, h, w, device="cuda").to(torch.bfloat16)
    xs = x.permute(0, 2, 3, 1).contiguous().requires_grad_(True)
    y = thin_conv.conv_to_1(xs, conv)
    dy = torch.randn(y.shape, device="cuda").to(torch.bfloat16)
    y.backward(dy)
    y_ref, dx_ref, dw_ref, db_ref = _ref(x, conv, 1.0, dy[:, None])
    _close(y[:, None], y_ref, 6e-3, "forward")
    _close(xs.grad.permute(0, 3, 1, 2), dx_ref, 6e-3, "dx")
    _close(conv.weight.grad, dw_ref, 2e-3, "dw")
    _close(conv.bias.grad, db_ref, 2e-3, "db")


def test_unsupported_shapes_are_refused(hip_lib_built):
    from octa_autosegmentation_amd import _native
    from octa_autosegmentation_amd.models import thin_conv
    assert not thin_conv.supported(torch.nn.Conv2d(1, 64, 3, 1, 1))
    assert not thin_conv.supported(torch.nn.Conv2d(1, 48, 7))
    assert not thin_conv.supported(torch.nn.Conv2d(1, 64, 4, 2, 1))
    conv = torch.nn.Conv2d(1, 48, 7).cuda()
    with pytest.raises(_native.OctaHipError):
        thin_conv.conv_from_1(torch.zeros(1, 16, 16, device="cuda"), conv)


def test_lrelu_backward_kernel_equals_the_torch_expression(hip_lib_built):
    """octa_lrelu_bwd_bf16 (one launch in the backward of the expand layer with its fused LeakyReLU) against torch.where(y > 0, dy, dy * slope),
    bit for bit, including a length that is no multiple of the 8-element vectors."""
    import ctypes
    from octa_autosegmentation_amd import _native
    g = torch.Generator(device="cuda").manual_seed(5)
    for n in (8 * 1000 + 5, 3, 64 * 303 * 303):
        y = torch.randn(n, device="cuda", generator=g).to(torch.bfloat16)
        y[::17] = 0.0
        dy = torch.randn(n, device="cuda", generator=g).to(torch.bfloat16)
        out = torch.empty_like(dy)
        _native.check(_native.lib().octa_lrelu_bwd_bf16(_native.ctx(0), ctypes.c_void_p(y.data_ptr()), ctypes.c_void_p(dy.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                                       n, 0.2, _native.current_stream_ptr()), "octa_lrelu_bwd_bf16")
        assert torch.equal(out, torch.where(y > 0, dy, dy * 0.2)), n
