"""csrc/blur.hip (reflection pad, blur-downsample, blur-upsample) against the torch ops the reference composes them
from (models/networks.py:244-289 and nn.ReflectionPad2d), forward and backward, both layouts and element types."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

SHAPES = [(2, 3, 8, 8), (1, 5, 7, 10), (2, 4, 2, 3), (1, 8, 33, 17), (1, 2, 304, 304),
          (2, 24, 9, 12)]      # NHWC bf16 with C % 8 == 0 runs the 8-channels-per-thread form (three groups here)


def _ref_down(x):
    c = x.shape[1]
    a = torch.tensor([1., 2., 1.], device=x.device, dtype=x.dtype)
    f = (a[:, None] * a[None, :] / 16)[None, None].repeat(c, 1, 1, 1)
    return nn.functional.conv2d(nn.functional.pad(x, [1, 1, 1, 1], mode="reflect"), f, stride=2, groups=c)


def _ref_up(x):
    c = x.shape[1]
    a = torch.tensor([1., 3., 3., 1.], device=x.device, dtype=x.dtype)
    f = (a[:, None] * a[None, :] / 64 * 4)[None, None].repeat(c, 1, 1, 1)
    y = nn.functional.conv_transpose2d(nn.functional.pad(x, [1, 1, 1, 1], mode="replicate"), f, stride=2, padding=2, groups=c)
    return y[:, :, 1:, 1:][:, :, :-1, :-1]


def _run(op, ref, shape, layout, dtype, **kw):
    from octa_autosegmentation_amd.models import resample
    torch.manual_seed(sum(shape))
    x = torch.randn(shape, device="cuda", dtype=torch.float64)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    g = torch.randn_like(yr)
    yr.backward(g)
    xin = x.to(dtype)
    gin = g.to(dtype)
    if layout == "nhwc":
        xin, gin = xin.permute(0, 2, 3, 1).contiguous(), gin.permute(0, 2, 3, 1).contiguous()
    xin.requires_grad_(True)
    y = op(xin, layout=layout, **kw)
    y.backward(gin)
    dx = xin.grad
    if layout == "nhwc":
        y, dx = y.permute(0, 3, 1, 2), dx.permute(0, 3, 1, 2)
    assert y.shape == yr.shape and y.dtype == dtype
    # fp32: same arithmetic up to summation order; bf16: one rounding of the result (2^-8 relative) on rounded inputs
    tol = 5e-6 if dtype == torch.float32 else 2e-2
    scale_y, scale_dx = yr.abs().max().item(), xr.grad.abs().max().item()
    assert (y.double() - yr).abs().max().item() <= tol * max(scale_y, 1.0)
    assert (dx.double() - xr.grad).abs().max().item() <= tol * max(scale_dx, 1.0)


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_blur_down_matches_torch(shape, layout, dtype):
    from octa_autosegmentation_amd.models import resample
    _run(resample.blur_down, _ref_down, shape, layout, dtype)


@pytest.mark.parametrize("shape", SHAPES + [(1, 2, 1, 1)])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_blur_up_matches_torch(shape, layout, dtype):
    from octa_autosegmentation_amd.models import resample
    _run(resample.blur_up, _ref_up, shape, layout, dtype)


@pytest.mark.parametrize("shape,pad", [((2, 3, 8, 8), 1), ((1, 5, 7, 10), 3), ((2, 4, 2, 3), 1), ((1, 8, 33, 17), 3), ((1, 2, 4, 4), 3), ((2, 24, 9, 12), 1),
                                       ((1, 2, 304, 304), 3)])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_reflect_pad_matches_torch(shape, pad, layout, dtype):
    from octa_autosegmentation_amd.models import resample
    _run(lambda x, layout: resample.reflect_pad(x, pad, layout), lambda x: nn.functional.pad(x, [pad] * 4, mode="reflect"), shape, layout, dtype)


def test_reflect_pad_forward_is_exact():
    from octa_autosegmentation_amd.models import resample
    x = torch.randn(2, 3, 19, 23, device="cuda")
    assert torch.equal(resample.reflect_pad(x, 3), nn.functional.pad(x, [3] * 4, mode="reflect"))
    xb = x.to(torch.bfloat16)
    assert torch.equal(resample.reflect_pad(xb, 1), nn.functional.pad(xb, [1] * 4, mode="reflect"))


def test_network_modules_use_the_kernels():
    """Downsample / Upsample / ReflectionPad2d modules of models/networks.py on a GPU tensor = their CPU (torch op) result."""
    from octa_autosegmentation_amd.models import networks as N
    torch.manual_seed(3)
    x = torch.randn(2, 6, 20, 28)
    for mod in (N.Downsample(6), N.Upsample(6), N.ReflectionPad2d(3)):
        want = mod(x)
        got = mod.cuda()(x.cuda())
        assert got.shape == want.shape
        assert (got.cpu() - want).abs().max().item() < 1e-5


def test_bad_arguments_are_rejected():
    from octa_autosegmentation_amd.models import resample
    with pytest.raises(RuntimeError):
        resample.reflect_pad(torch.randn(1, 1, 3, 3, device="cuda"), 3)
