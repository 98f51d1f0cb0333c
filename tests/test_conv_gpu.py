"""GPU parity of the hand-written MFMA 3x3 convolution (csrc/conv.hip) against a plain torch fp32 reference of the
same op on the same bf16-rounded operands. Tolerance: the kernel accumulates in fp32 and rounds the result to
bf16 once (relative 2^-9), the fp32 reference differs from it by accumulation order only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ref_conv(x_nhwc, w, stride):
    import torch
    import torch.nn.functional as F
    y = F.conv2d(x_nhwc.float().permute(0, 3, 1, 2), w.float(), stride=stride, padding=1)
    return y.permute(0, 2, 3, 1).contiguous()


def _check(got, want):
    import torch
    err = (got.float() - want).abs()
    scale = want.abs().max().item() + 1e-6
    # bf16 output rounding: half an ulp = 2^-9 relative to the value; allow 2^-8 of the tensor scale
    assert err.max().item() <= scale * 2.0 ** -8 + 1e-6, (err.max().item(), scale)


@pytest.mark.parametrize("n,h,w,cin,cout,stride", [
    (2, 40, 70, 32, 32, 1), (1, 16, 64, 64, 64, 1), (1, 37, 45, 64, 32, 1), (2, 24, 40, 128, 64, 1),
    (1, 48, 80, 32, 64, 2), (1, 38, 66, 64, 128, 2), (1, 8, 32, 256, 256, 1),
])
def test_forward_matches_torch(hip_lib_built, n, h, w, cin, cout, stride):
    import torch
    from octa_autosegmentation_amd.models import mfma_conv
    g = torch.Generator(device="cuda").manual_seed(h * 1000 + cin)
    x = torch.randn(n, h, w, cin, device="cuda", generator=g).to(torch.bfloat16)
    wt = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (3.0 * cin ** 0.5)).to(torch.bfloat16)
    # asymmetric weights: a transposed tap or channel order cannot pass
    got = mfma_conv.conv3x3_nhwc(x, mfma_conv.pack_weight(wt), stride=stride)
    want = _ref_conv(x, wt, stride)
    assert got.shape == want.shape
    _check(got, want)


@pytest.mark.parametrize("n,h,w,cin,cout,stride", [(1, 24, 40, 32, 64, 1), (2, 16, 32, 64, 32, 1), (1, 48, 64, 32, 64, 2), (1, 20, 36, 64, 64, 2)])
def test_data_gradient_matches_torch(hip_lib_built, n, h, w, cin, cout, stride):
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.models import mfma_conv
    g = torch.Generator(device="cuda").manual_seed(7 * h + cout)
    x = torch.randn(n, cin, h, w, device="cuda", generator=g).to(torch.bfloat16).float().requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (3.0 * cin ** 0.5)).to(torch.bfloat16)
    y = F.conv2d(x, wt.float(), stride=stride, padding=1)
    dy = torch.randn(y.shape, device="cuda", generator=g).to(torch.bfloat16)
    y.backward(dy.float())
    want = x.grad.permute(0, 2, 3, 1).contiguous()
    dy_nhwc = dy.permute(0, 2, 3, 1).contiguous()
    got = mfma_conv.conv3x3_nhwc(dy_nhwc, mfma_conv.pack_weight_dgrad(wt), stride=1, in_dilation=stride)
    assert got.shape == want.shape
    _check(got, want)


@pytest.mark.parametrize("n,h,w,cin,cout", [(1, 16, 32, 32, 32), (2, 24, 40, 64, 64), (1, 37, 45, 64, 32), (2, 19, 70, 32, 64), (1, 8, 32, 128, 128)])
def test_weight_gradient_matches_torch(hip_lib_built, n, h, w, cin, cout):
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.models import mfma_conv
    g = torch.Generator(device="cuda").manual_seed(3 * h + cin)
    x = torch.randn(n, cin, h, w, device="cuda", generator=g).to(torch.bfloat16)
    wt = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (3.0 * cin ** 0.5)).float().requires_grad_(True)
    y = F.conv2d(x.float(), wt, stride=1, padding=1)
    dy = torch.randn(y.shape, device="cuda", generator=g).to(torch.bfloat16)
    y.backward(dy.float())
    got = mfma_conv.conv3x3_nhwc_wgrad(x.permute(0, 2, 3, 1).contiguous(), dy.permute(0, 2, 3, 1).contiguous())
    want = wt.grad
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    scale = want.abs().max().item()
    assert err <= scale * 1e-4 + 1e-5, (err, scale)   # fp32 accumulation on exact bf16 products: order only
