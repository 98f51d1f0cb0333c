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
    (1, 213, 37, 32, 64, 1), (2, 200, 64, 64, 128, 1),     # >= 200 output rows: the 16-row-tile variant of the DMA-staged kernel
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


@pytest.mark.parametrize("n,h,w,cin,cout,stride", [(1, 24, 40, 32, 64, 1), (2, 16, 32, 64, 32, 1), (1, 48, 64, 32, 64, 2), (1, 20, 36, 64, 64, 2),
                                                   (1, 207, 40, 64, 64, 1), (1, 104, 36, 64, 64, 2)])
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


@pytest.mark.parametrize("n,h,w,cin,cout", [(2, 76, 76, 64, 64), (1, 37, 45, 32, 64), (1, 19, 23, 64, 32), (1, 210, 40, 64, 64), (1, 8, 9, 128, 128)])
def test_reflection_fused_convolution_matches_torch(hip_lib_built, n, h, w, cin, cout):
    """ReflectionPad2d(1) + Conv2d(3, padding 0) of the generator's ResNet blocks (models/networks.py:151-176) with the reflection
    fused into the halo fetch of the forward and weight-gradient kernels (octa_conv3x3_nhwc_fwd_pad / _wgrad_pad) and the data gradient
    as the full convolution folded by the reflection's adjoint: output, input gradient and weight gradient against torch fp32 on the
    same bf16-rounded operands; and against the unfused composition (padded copy + zero-padded kernel + crop)."""
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.models import mfma_conv
    g = torch.Generator(device="cuda").manual_seed(11 * h + cin)
    x0 = torch.randn(n, h, w, cin, device="cuda", generator=g).to(torch.bfloat16)
    wt0 = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (3.0 * cin ** 0.5)).to(torch.bfloat16)
    dy = torch.randn(n, h, w, cout, device="cuda", generator=g).to(torch.bfloat16)
    xr = x0.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = wt0.float().requires_grad_(True)
    yr = F.conv2d(F.pad(xr, (1, 1, 1, 1), mode="reflect"), wr)
    yr.backward(dy.float().permute(0, 3, 1, 2))
    x = x0.clone().requires_grad_(True)
    wt = wt0.float().requires_grad_(True)
    y = mfma_conv.conv3x3_reflect(x, wt)
    assert y.shape == (n, h, w, cout)
    y.backward(dy)
    _check(y.detach(), yr.detach().permute(0, 2, 3, 1))
    # the border pixels' gradient is rounded to bf16 twice (the full convolution's output, then the fold): 2^-7 of the tensor scale
    want_dx = xr.grad.permute(0, 2, 3, 1)
    assert (x.grad.float() - want_dx).abs().max().item() <= want_dx.abs().max().item() * 2.0 ** -7, ((x.grad.float() - want_dx).abs().max().item(), want_dx.abs().max().item())
    assert (x.grad.float() - want_dx)[:, 2:-2, 2:-2].abs().max().item() <= want_dx.abs().max().item() * 2.0 ** -8 + 1e-6      # interior: one rounding
    rel = (wt.grad - wr.grad).abs().max().item() / wr.grad.abs().max().item()
    assert rel < 2e-3, rel
    # the unfused composition gives the same bf16 values up to the summation order
    old, mfma_conv.USE_FUSED_REFLECT = mfma_conv.USE_FUSED_REFLECT, False
    try:
        y2 = mfma_conv.conv3x3_reflect(x0, wt0.float())
    finally:
        mfma_conv.USE_FUSED_REFLECT = old
    assert (y2.float() - y.detach().float()).abs().max().item() <= 2.0 ** -6 * yr.abs().max().item()
    # statistics from the epilogue (the generator's residual blocks: reflect-padded convolution -> InstanceNorm): the same result, and the
    # slots hold the sums of the ROUNDED values per image and channel; the norm fed with them equals the norm that makes its own pass
    y3, part = mfma_conv.conv3x3_reflect(x0, wt0.float(), True)
    assert torch.equal(y3, y.detach())
    tot, yf = part.sum(0), y3.double()
    scale = yr.abs().max().item()
    assert torch.allclose(tot[..., 0], yf.sum((1, 2)), rtol=1e-5, atol=1e-3 * scale)
    assert torch.allclose(tot[..., 1], (yf * yf).sum((1, 2)), rtol=1e-5, atol=1e-3 * scale * scale)
    z_slots = mfma_conv.instance_norm_leaky_relu_nhwc(y3, None, None, 0.0, 1e-5, part)
    z_pass = mfma_conv.instance_norm_leaky_relu_nhwc(y3, None, None, 0.0, 1e-5)
    assert (z_slots.float() - z_pass.float()).abs().max().item() <= 2.0 ** -7 * z_pass.float().abs().max().item()


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


@pytest.mark.parametrize("stride,cin,cout,h,w", [(1, 32, 64, 24, 40), (2, 32, 64, 24, 40), (2, 64, 128, 16, 48), (1, 1, 32, 20, 36)])
def test_autograd_function_gradients(hip_lib_built, stride, cin, cout, h, w):
    """conv3x3 autograd binding (incl. channel padding of a 1-channel input and the stride-2 weight gradient through
    the four parity planes) against torch autograd on the same bf16-rounded operands."""
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.models import mfma_conv
    g = torch.Generator(device="cuda").manual_seed(11 * h + cout)
    x = torch.randn(2, cin, h, w, device="cuda", generator=g).to(torch.bfloat16)
    wt = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (3.0 * cin ** 0.5)).to(torch.bfloat16).float()
    xr = x.float().requires_grad_(True)
    wr = wt.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, stride=stride, padding=1)
    dy = torch.randn(yr.shape, device="cuda", generator=g).to(torch.bfloat16)
    yr.backward(dy.float())
    xm = x.permute(0, 2, 3, 1).contiguous().requires_grad_(cin % 32 == 0)
    wm = wt.clone().requires_grad_(True)
    ym = mfma_conv.conv3x3(xm, wm, stride)
    ym.backward(dy.permute(0, 2, 3, 1).contiguous())
    _check(ym.detach(), yr.detach().permute(0, 2, 3, 1))
    err = (wm.grad - wr.grad).abs().max().item()
    assert err <= wr.grad.abs().max().item() * 1e-3 + 1e-5, err
    if cin % 32 == 0:
        _check(xm.grad, xr.grad.permute(0, 2, 3, 1))


@pytest.mark.parametrize("c1,c2,cout", [(32, 32, 32), (64, 64, 64), (128, 64, 96)])
def test_virtual_concat_conv(hip_lib_built, c1, c2, cout):
    """conv over cat(x1, x2) without the cat: forward, both data gradients, weight gradient vs torch autograd."""
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.models import mfma_conv
    g = torch.Generator(device="cuda").manual_seed(c1 + cout)
    x1 = torch.randn(2, 19, 40, c1, device="cuda", generator=g).to(torch.bfloat16)
    x2 = torch.randn(2, 19, 40, c2, device="cuda", generator=g).to(torch.bfloat16)
    wt = (torch.randn(cout, c1 + c2, 3, 3, device="cuda", generator=g) / (3.0 * (c1 + c2) ** 0.5)).to(torch.bfloat16).float()
    x1r, x2r, wr = x1.float().requires_grad_(True), x2.float().requires_grad_(True), wt.clone().requires_grad_(True)
    yr = F.conv2d(torch.cat((x1r, x2r), -1).permute(0, 3, 1, 2), wr, padding=1).permute(0, 2, 3, 1)
    dy = torch.randn(yr.shape, device="cuda", generator=g).to(torch.bfloat16)
    yr.backward(dy.float())
    x1m, x2m, wm = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    ym = mfma_conv.conv3x3_cat(x1m, x2m, wm)
    ym.backward(dy)
    _check(ym.detach(), yr.detach())
    _check(x1m.grad, x1r.grad)
    _check(x2m.grad, x2r.grad)
    assert (wm.grad - wr.grad).abs().max().item() <= wr.grad.abs().max().item() * 1e-3 + 1e-5


@pytest.mark.parametrize("cin,cout", [(64, 32), (128, 64)])
def test_transposed_conv_2x2(hip_lib_built, cin, cout):
    """ConvTranspose2d(k=2, s=2) through the adjoint-of-stride-2-conv formulation vs torch."""
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.models import mfma_conv
    g = torch.Generator(device="cuda").manual_seed(cin)
    x = torch.randn(2, 12, 20, cin, device="cuda", generator=g).to(torch.bfloat16)
    wt = (torch.randn(cin, cout, 2, 2, device="cuda", generator=g) / cin ** 0.5).to(torch.bfloat16).float()
    xr, wr = x.float().requires_grad_(True), wt.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr.permute(0, 3, 1, 2), wr, stride=2).permute(0, 2, 3, 1)
    dy = torch.randn(yr.shape, device="cuda", generator=g).to(torch.bfloat16)
    yr.backward(dy.float())
    xm, wm = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    ym = mfma_conv.conv_transpose_kxk_nhwc(xm, wm, 2)
    assert ym.shape == yr.shape
    ym.backward(dy)
    _check(ym.detach(), yr.detach())
    _check(xm.grad, xr.grad)
    assert (wm.grad - wr.grad).abs().max().item() <= wr.grad.abs().max().item() * 1e-3 + 1e-5


def test_parity_scatter_forms(hip_lib_built):
    """The zero-insertion-free forms (one scattered launch per output parity) agree with torch too."""
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.models import mfma_conv
    g = torch.Generator(device="cuda").manual_seed(9)
    w = (torch.randn(64, 32, 3, 3, device="cuda", generator=g) / 17.0).to(torch.bfloat16).float()
    x = torch.randn(1, 32, 24, 40, device="cuda", generator=g).to(torch.bfloat16).float().requires_grad_(True)
    y = F.conv2d(x, w, stride=2, padding=1)
    dy = torch.randn(y.shape, device="cuda", generator=g).to(torch.bfloat16)
    y.backward(dy.float())
    got = mfma_conv.conv3x3_s2_dgrad(dy.permute(0, 2, 3, 1).contiguous(), w)
    _check(got, x.grad.permute(0, 2, 3, 1))
    wt = (torch.randn(64, 32, 2, 2, device="cuda", generator=g) / 8.0).to(torch.bfloat16).float()
    xt = torch.randn(2, 10, 18, 64, device="cuda", generator=g).to(torch.bfloat16)
    want = F.conv_transpose2d(xt.float().permute(0, 3, 1, 2), wt, stride=2).permute(0, 2, 3, 1)
    _check(mfma_conv.conv_transpose_2x2_fwd(xt, wt), want)


def test_normalise_on_load_equals_materialised(hip_lib_built):
    """conv3x3_lazy (InstanceNorm + LeakyReLU applied while the kernels load their input) matches the
    materialised route to the last bf16 bit in the forward pass and in all gradients; whole network with networks.USE_LAZY_NORM too."""
    import torch
    from octa_autosegmentation_amd.models import mfma_conv as mc, networks
    g = torch.Generator(device="cuda").manual_seed(21)
    raw = torch.randn(2, 20, 36, 64, device="cuda", generator=g).to(torch.bfloat16)
    skip = torch.randn(2, 20, 36, 32, device="cuda", generator=g).to(torch.bfloat16)
    gam, bet = torch.rand(64, device="cuda", generator=g) + 0.5, torch.randn(64, device="cuda", generator=g) * 0.1
    w = (torch.randn(64, 96, 3, 3, device="cuda", generator=g) / 29.0)
    dy = torch.randn(2, 20, 36, 64, device="cuda", generator=g).to(torch.bfloat16)
    outs = []
    for lazy in (False, True):
        r, gm, bt, wt, sk = raw.clone().requires_grad_(True), gam.clone().requires_grad_(True), bet.clone().requires_grad_(True), \
            w.clone().requires_grad_(True), skip.clone().requires_grad_(True)
        if lazy:
            y = mc.conv3x3_lazy(mc.lazy_norm(r, gm, bt), wt, 1, (sk, None, None))
        else:
            y = mc.conv3x3_cat(mc.instance_norm_leaky_relu_nhwc(r, gm, bt), sk, wt)
        y.backward(dy)
        outs.append((y.detach(), r.grad, sk.grad, gm.grad, bt.grad, wt.grad))
    for a_, b_ in zip(*outs):
        if a_.dtype == torch.bfloat16:
            # same arithmetic, but the materialised route runs the DMA-staged kernel (16-channel slices) and the lazy route
            # the register-staged one (32-channel slices): fp32 partial sums meet in a different order -> last bf16 bit
            assert (a_.float() - b_.float()).abs().max().item() <= a_.float().abs().max().item() * 2 ** -7
        else:
            assert (a_ - b_).abs().max().item() <= a_.abs().max().item() * 1e-3   # fp32 atomics: arrival order only
    torch.manual_seed(5)
    net = networks.DynUNet(2, 1, 1, [3, 3, 3, 3, 3], [1, 2, 2, 2, 1], [1, 2, 2, 2, 1]).cuda()
    networks.init_weights(net, "kaiming")
    x = torch.rand(1, 1, 32, 64, device="cuda")
    res = []
    for lazy in (False, True):
        networks.USE_LAZY_NORM = lazy
        try:
            with torch.autocast("cuda", dtype=torch.bfloat16), torch.no_grad():
                res.append(net(x).float())
        finally:
            networks.USE_LAZY_NORM = False
    # 20 bf16 layers with instance norms over as few as 32 pixels: a last-bit difference per layer (different slice depth of
    # the two kernels) grows to a few percent of the logit range; a wrong scale / shift would be off by O(1)
    assert (res[0] - res[1]).abs().max().item() <= 0.05 * max(res[0].abs().max().item(), 1.0)
    assert (res[0] - res[1]).abs().mean().item() <= 0.01 * max(res[0].abs().max().item(), 1.0)


def test_head_kernels(hip_lib_built):
    import torch
    from octa_autosegmentation_amd.models import mfma_conv
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(2, 37, 53, 32, device="cuda", generator=g).to(torch.bfloat16)
    w = torch.randn(1, 32, 1, 1, device="cuda", generator=g) * 0.2
    b = torch.tensor([0.3], device="cuda")
    xr, wr, br = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = (xr * wr.view(1, 1, 1, 32)).sum(-1, keepdim=True) + br
    dy = torch.randn(yr.shape, device="cuda", generator=g).to(torch.bfloat16)
    yr.backward(dy.float())
    xm, wm, bm = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ym = mfma_conv.conv1x1_bias_nhwc(xm, wm, bm)
    ym.backward(dy)
    _check(ym.detach(), yr.detach())
    _check(xm.grad, xr.grad)
    assert (wm.grad - wr.grad).abs().max().item() <= wr.grad.abs().max().item() * 1e-3
    assert abs(bm.grad.item() - br.grad.item()) <= abs(br.grad.item()) * 1e-3 + 1e-3


def test_dynunet_mfma_path_matches_torch_reference(hip_lib_built):
    """Whole DynUNet-S, forward logits and parameter gradients: channels-last bf16 path on the hand-written kernels
    against the plain torch fp32 modules with the same parameters. Activations and activation gradients are bf16
    end to end, exactly as under torch's own bf16 autocast, so the yardstick is torch autocast's own distance from
    the fp32 reference on the same problem: logits within 1.5x of it, every parameter gradient at least as well
    aligned with the fp32 gradient (cosine) as autocast's minus 0.1 (this path also keeps the activation GRADIENTS in
    bf16 between all ops, autocast keeps them fp32 through the norms; measured 0.82 vs 0.88 on the worst tensor, the
    first block's norm bias, and 0.99+ near the head)."""
    import torch
    from octa_autosegmentation_amd.models import networks
    torch.manual_seed(3)
    net = networks.DynUNet(2, 1, 1, [3, 3, 3, 3, 3], [1, 2, 2, 2, 1], [1, 2, 2, 2, 1]).cuda()
    networks.init_weights(net, "kaiming")
    x = torch.rand(2, 1, 64, 96, device="cuda")
    tgt = (torch.rand(2, 1, 64, 96, device="cuda") > 0.7).float()

    def run(mfma, autocast=False):
        old = (networks.USE_MFMA_CONV, networks.USE_FUSED_NORM)
        networks.USE_MFMA_CONV, networks.USE_FUSED_NORM = mfma, mfma
        try:
            net.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast or mfma):   # the MFMA path is the bf16 path
                out = net(x)
            loss = torch.nn.functional.binary_cross_entropy_with_logits(out.float(), tgt)
            loss.backward()
            return out.float().detach(), {k: p.grad.float().clone() for k, p in net.named_parameters()}
        finally:
            networks.USE_MFMA_CONV, networks.USE_FUSED_NORM = old

    ref_out, ref_g = run(False)
    ac_out, ac_g = run(False, autocast=True)
    got_out, got_g = run(True)
    assert got_out.shape == ref_out.shape
    e_got, e_ac = (got_out - ref_out).abs().max().item(), (ac_out - ref_out).abs().max().item()
    scale = (ref_out.max() - ref_out.min()).item()
    assert e_got <= 1.5 * e_ac + 0.005 * scale, (e_got, e_ac, scale)

    def cos(u, v):
        return (torch.dot(u, v) / (u.norm() * v.norm() + 1e-30)).item()

    for k in ref_g:
        a, b, c = got_g[k].flatten(), ref_g[k].flatten(), ac_g[k].flatten()
        if b.norm().item() < 1e-12:
            continue
        assert cos(a, b) >= cos(c, b) - 0.1 and cos(a, b) > 0.75, (k, cos(a, b), cos(c, b))
        dev_ac = abs(c.norm().item() / b.norm().item() - 1.0)
        assert abs(a.norm().item() / b.norm().item() - 1.0) < dev_ac + 0.3, (k, a.norm().item(), c.norm().item(), b.norm().item())


def test_parity_fused_up_convolution_equals_the_zero_insertion_form(hip_lib_built):
    """octa_conv3x3_s2t_nhwc (round 5: the data gradient of a stride-2 layer and the 2x2 transposed convolution with the four output
    parities fused) against the zero-insertion form of rounds 1-4 on the same packed weights -- the same sums in another order (fp32
    accumulation, one bf16 rounding) -- and, through the autograd bindings, against torch fp32. Odd small sizes, both channel widths,
    the residual, the one-tap-per-class mask of the transposed convolution."""
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.models import mfma_conv as mc
    g = torch.Generator(device="cuda").manual_seed(77)
    for (n, h, w, cin, cout, mask) in ((2, 19, 37, 64, 32, 0x1ff), (1, 8, 32, 32, 64, 0x1ff), (2, 33, 70, 128, 64, 0x1ff), (1, 76, 76, 256, 128, 0b000011011),
                                       (2, 5, 3, 64, 32, 0b000011011), (1, 152, 152, 64, 32, 0x1ff)):
        x = torch.randn(n, h, w, cin, device="cuda", generator=g).to(torch.bfloat16)
        wt = (torch.randn(9, cout, cin, device="cuda", generator=g) / (3.0 * cin ** 0.5)).to(torch.bfloat16)
        if mask != 0x1ff:
            keep = torch.tensor([(mask >> t) & 1 for t in range(9)], device="cuda", dtype=torch.bfloat16)
            wt = wt * keep[:, None, None]
        wt = mc.slice_major(wt)                  # nominal [9][Cout][Cin] -> the kernels' storage order (the mask names NOMINAL taps)
        res = torch.randn(n, 2 * h, 2 * w, cout, device="cuda", generator=g).to(torch.bfloat16)
        for r in (None, res):
            a = mc.conv3x3_s2t_nhwc(x, wt, mask, r)
            if r is None:
                b = mc.conv3x3_nhwc(x, wt, stride=1, in_dilation=2, tap_mask=mask)
            else:
                b = mc.conv3x3_nhwc(x, wt, stride=1, in_dilation=2, tap_mask=0x1ff, residual=r)      # the residual epilogue lives in the unmasked kernel (masked taps are zero)
            assert a.shape == b.shape == (n, 2 * h, 2 * w, cout)
            scale = b.float().abs().max().item()
            assert (a.float() - b.float()).abs().max().item() <= 2.0 ** -7 * scale, (n, h, w, cin, cout, mask)
            assert (a != b).float().mean().item() < 0.02           # summation order only: a few values round the other way
    # through autograd: stride-2 layer and transposed convolution against torch fp32 on the same bf16-rounded operands
    x = torch.randn(2, 40, 64, 32, device="cuda", generator=g).to(torch.bfloat16)
    w2 = (torch.randn(64, 32, 3, 3, device="cuda", generator=g) / 17.0).to(torch.bfloat16).float()
    xm, wm = x.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    ym = mc.conv3x3(xm, wm, 2)
    dy = torch.randn(ym.shape, device="cuda", generator=g).to(torch.bfloat16)
    ym.backward(dy)
    xr, wr = x.float().permute(0, 3, 1, 2).requires_grad_(True), w2.clone().requires_grad_(True)
    F.conv2d(xr, wr, stride=2, padding=1).backward(dy.float().permute(0, 3, 1, 2))
    assert (xm.grad.float() - xr.grad.permute(0, 2, 3, 1)).abs().max().item() <= 2.0 ** -8 * xr.grad.abs().max().item() + 1e-6
    wt2 = (torch.randn(64, 32, 2, 2, device="cuda", generator=g) / 11.0).to(torch.bfloat16).float()
    xm, wm = torch.randn(2, 21, 30, 64, device="cuda", generator=g).to(torch.bfloat16).requires_grad_(True), wt2.clone().requires_grad_(True)
    ym = mc.conv_transpose_kxk_nhwc(xm, wm, 2)
    yr = F.conv_transpose2d(xm.detach().float().permute(0, 3, 1, 2), wt2, stride=2)
    assert (ym.float() - yr.permute(0, 2, 3, 1)).abs().max().item() <= 2.0 ** -8 * yr.abs().max().item() + 1e-6


def test_epilogue_statistics_feed_the_norm(hip_lib_built):
    """InstanceNorm statistics accumulated in the convolution's epilogue give the same normalised tensor as the
    statistics pass over the stored result (same bf16 values, fp32 / double sums in a different order)."""
    import torch
    from octa_autosegmentation_amd.models import mfma_conv as mc
    g = torch.Generator(device="cuda").manual_seed(31)
    # (203, 70) with 64 output channels takes the 16-row tiles, which write two 8-row slots of the partials layout each (the last tile
    # row has a lower half outside the image); (216, 64): an even number of 8-row tiles
    for (h, w, cin, cout, st) in ((37, 45, 32, 64, 1), (24, 40, 64, 32, 2), (203, 70, 32, 64, 1), (216, 64, 64, 128, 1)):
        x = torch.randn(2, h, w, cin, device="cuda", generator=g).to(torch.bfloat16)
        wt = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (3.0 * cin ** 0.5)
        gam, bet = torch.rand(cout, device="cuda", generator=g) + 0.5, torch.randn(cout, device="cuda", generator=g) * 0.1
        for form in ("slots", "tiles"):         # round 5: double slots + atomics (the default form); round 1: per-tile float partials
            y, part = mc.conv3x3(x, wt, st, form)
            assert part.dtype == (torch.float64 if form == "slots" else torch.float32)
            a = mc.instance_norm_leaky_relu_nhwc(y, gam, bet, partials=part)
            b = mc.instance_norm_leaky_relu_nhwc(y, gam, bet)
            assert (a.float() - b.float()).abs().max().item() <= 2.0 ** -7 * b.float().abs().max().item()
            assert (a != b).float().mean().item() < 0.01      # a handful of values may round the other way
            if form == "slots":
                # the slots hold exactly the sums of the stored bf16 values (fp32 within a tile, double across tiles)
                yf = y.double()
                want = torch.stack((yf.sum(dim=(1, 2)), (yf * yf).sum(dim=(1, 2))), dim=-1)            # [N][C][2]
                got = part.sum(dim=0)
                assert (got - want).abs().max().item() <= 1e-5 * want.abs().max().item(), (h, w, cin, cout, st)
    # two virtually concatenated inputs (the decoder's convolution) with statistics
    x1 = torch.randn(2, 40, 72, 64, device="cuda", generator=g).to(torch.bfloat16)
    x2 = torch.randn(2, 40, 72, 32, device="cuda", generator=g).to(torch.bfloat16)
    wt = torch.randn(32, 96, 3, 3, device="cuda", generator=g) / (3.0 * 96 ** 0.5)
    y, part = mc.conv3x3_cat(x1, x2, wt, True)
    yf = y.double()
    want = torch.stack((yf.sum(dim=(1, 2)), (yf * yf).sum(dim=(1, 2))), dim=-1)
    assert (part.sum(dim=0) - want).abs().max().item() <= 1e-5 * want.abs().max().item()
    assert torch.equal(y, mc.conv3x3_cat(x1, x2, wt))                 # the statistics epilogue does not change the stored result


def test_weight_pack_plan_is_bit_exact_and_follows_updates(hip_lib_built):
    """One-launch packing of all KxK weights (octa_pack_conv_weights) against the torch formulation of the same layouts:
    3x3 with a channel-padded input, 4x4, the 2x2 transposed convolution; bf16 bits identical; refreshed after an in-place
    optimiser-style update."""
    import torch
    from octa_autosegmentation_amd.models import mfma_conv as mc
    g = torch.Generator(device="cuda").manual_seed(5)
    mk = lambda *s: torch.nn.Parameter(torch.randn(*s, device="cuda", generator=g))
    w3, w3p, w4, wt = mk(64, 32, 3, 3), mk(32, 1, 3, 3), mk(128, 64, 4, 4), mk(64, 32, 2, 2)
    plan = mc.WeightPackPlan([w3, w3p, w4], [wt])

    def ref():
        mc.USE_PACK_PLAN = False
        try:
            return [mc.pack_weight(w3), mc.pack_weight_dgrad(w3), mc.pack_weight(w3p, 32), mc.pack_weight_dgrad(w3p, 32),
                    mc.pack_weight(w4), mc.pack_weight_dgrad(w4), *mc.pack_convt2x2(wt)]
        finally:
            mc.USE_PACK_PLAN = True

    def got():
        return [mc.pack_weight(w3), mc.pack_weight_dgrad(w3), mc.pack_weight(w3p, 32), mc.pack_weight_dgrad(w3p, 32),
                mc.pack_weight(w4), mc.pack_weight_dgrad(w4), *mc.pack_convt2x2(wt)]

    for a, b in zip(got(), ref()):
        assert a.shape == b.shape and a.dtype == torch.bfloat16
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    assert mc.pack_weight(w3).data_ptr() == plan.fwd[0].data_ptr()          # served from the plan, not recomputed
    with torch.no_grad():
        for w in (w3, w3p, w4, wt):
            w.add_(0.25)
    for a, b in zip(got(), ref()):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    # an unregistered tensor takes the torch formulation
    other = torch.randn(32, 32, 3, 3, device="cuda")
    assert mc.pack_weight(other).shape == (9, 32, 32)


@pytest.mark.parametrize("hw", [(19, 23), (64, 96)])
def test_transposed_conv_1x1_weight_gradient_on_the_mfma_kernel(hip_lib_built, hw):
    """ConvTranspose2d(k=1, s=1): GEMM forward / data gradient; weight gradient as a hand-split batched GEMM with fp32 results
    (pixel counts divisible by the split: 64 x 96) or through the tap-masked MFMA kernel (19 x 23)."""
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.models import mfma_conv
    cin, cout = 128, 64
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(2, hw[0], hw[1], cin, device="cuda", generator=g).to(torch.bfloat16)
    wt = (torch.randn(cin, cout, 1, 1, device="cuda", generator=g) / cin ** 0.5).to(torch.bfloat16).float()
    xr, wr = x.float().requires_grad_(True), wt.clone().requires_grad_(True)
    yr = F.conv_transpose2d(xr.permute(0, 3, 1, 2), wr, stride=1).permute(0, 2, 3, 1)
    dy = torch.randn(yr.shape, device="cuda", generator=g).to(torch.bfloat16)
    yr.backward(dy.float())
    xm, wm = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
    ym = mfma_conv.conv_transpose_kxk_nhwc(xm, wm, 1)
    assert ym.shape == yr.shape
    ym.backward(dy)
    _check(ym.detach(), yr.detach())
    _check(xm.grad, xr.grad)
    assert wm.grad.shape == wr.grad.shape
    assert (wm.grad - wr.grad).abs().max().item() <= wr.grad.abs().max().item() * 1e-3 + 1e-5


@pytest.mark.parametrize("stride_of_layer", [1, 2])
def test_residual_epilogue_equals_separate_bf16_addition(hip_lib_built, stride_of_layer):
    """octa_conv3x3_nhwc_fwd6: result + residual in the epilogue is bit for bit conv -> bf16, then a bf16 tensor addition
    (the data gradient of a stride-1 / stride-2 layer is the dilation-1 / dilation-2 launch)."""
    import torch
    from octa_autosegmentation_amd.models import mfma_conv
    g = torch.Generator(device="cuda").manual_seed(77 + stride_of_layer)
    cin, cout, n, h, w = 64, 32, 2, 21, 35
    dy = torch.randn(n, h, w, cin, device="cuda", generator=g).to(torch.bfloat16)
    wt = (torch.randn(cin, cout, 3, 3, device="cuda", generator=g) / (3.0 * cin ** 0.5)).to(torch.bfloat16)
    wd = mfma_conv.pack_weight_dgrad(wt)          # [9, cout, cin]: conv(dy, wd) has `cout` channels
    plain = mfma_conv.conv3x3_nhwc(dy, wd, stride=1, in_dilation=stride_of_layer)
    res = torch.randn(plain.shape, device="cuda", generator=g).to(torch.bfloat16)
    fused = mfma_conv.conv3x3_nhwc(dy, wd, stride=1, in_dilation=stride_of_layer, residual=res)
    assert torch.equal(fused.view(torch.int16), (plain + res).view(torch.int16))


def test_skip_gradient_fusion_leaves_the_gradients_unchanged(hip_lib_built):
    """DynUNet-S step with the decoder's skip gradients riding in the encoder's data-gradient epilogue vs autograd's own
    additions: every parameter gradient identical up to the fp32 atomics' arrival order of the weight-gradient kernels."""
    import torch
    from octa_autosegmentation_amd.models import mfma_conv
    from octa_autosegmentation_amd.models.networks import DynUNet, init_weights
    torch.manual_seed(3)
    net = DynUNet(spatial_dims=2, in_channels=1, out_channels=1, kernel_size=[3, 3, 3, 3, 3], strides=[1, 2, 2, 2, 1],
                  upsample_kernel_size=[1, 2, 2, 2, 1]).cuda()
    init_weights(net, "kaiming")
    x = torch.rand(2, 1, 64, 96, device="cuda")
    grads = []
    for fused in (True, False):
        mfma_conv.USE_SKIP_GRAD_FUSION = fused
        try:
            net.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = net(x)
            out.float().square().mean().backward()
            grads.append([p.grad.clone() for p in net.parameters()])
        finally:
            mfma_conv.USE_SKIP_GRAD_FUSION = True
    for a, b in zip(*grads):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-6 + 1e-4 * b.abs().max().item())


@pytest.mark.parametrize("c,h,w", [(32, 37, 53), (64, 16, 24)])
def test_fused_norm_head_matches_torch_fp64(hip_lib_built, c, h, w):
    """InstanceNorm(affine) + LeakyReLU + 1x1 head to one channel in one pair of passes (csrc/norm.hip) against the plain torch
    float64 formulation (MIOpen's fp32 instance-norm backward is itself off by 1e-2 on odd plane sizes) on the same bf16 input: logits within one bf16 rounding, gradients within 2^-7 of their scale (the fused
    route rounds LESS than the unfused one: the normalised tensor and its gradient are never rounded to bf16)."""
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.models import mfma_conv
    g = torch.Generator(device="cuda").manual_seed(c + h)
    n = 2
    x = (torch.randn(n, h, w, c, device="cuda", generator=g) * 1.7 + 0.3).to(torch.bfloat16)
    gamma = torch.rand(c, device="cuda", generator=g) + 0.5
    beta = torch.randn(c, device="cuda", generator=g) * 0.2
    hw_ = torch.randn(1, c, 1, 1, device="cuda", generator=g) / c ** 0.5
    hb = torch.randn(1, device="cuda", generator=g)
    dl = torch.randn(n, h, w, 1, device="cuda", generator=g).to(torch.bfloat16)
    # reference
    xr, gr, br, wr, hbr = (t.clone().double().requires_grad_(True) for t in (x, gamma, beta, hw_, hb))
    y = F.leaky_relu(F.instance_norm(xr.permute(0, 3, 1, 2), weight=gr, bias=br, eps=1e-5), 0.01)
    lr = (y.permute(0, 2, 3, 1) @ wr.reshape(-1, 1) + hbr)
    lr.backward(dl.double())
    # fused
    xm, gm, bm, wm, hbm = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), \
        hw_.clone().requires_grad_(True), hb.clone().requires_grad_(True)
    lm = mfma_conv.instance_norm_leaky_relu_head1_nhwc(xm, gm, bm, 0.01, 1e-5, wm, hbm)
    assert lm.shape == lr.shape and lm.dtype == torch.bfloat16
    lm.backward(dl)
    _check(lm.detach(), lr.detach().float())

    def close(a, b, rel):
        assert a.shape == b.shape
        assert (a.float() - b.float()).abs().max().item() <= b.abs().max().item() * rel + 1e-6, ((a.float() - b.float()).abs().max().item(), b.abs().max().item())
    close(xm.grad, xr.grad, 2.0 ** -7)
    close(gm.grad, gr.grad, 2e-3)
    close(bm.grad, br.grad, 2e-3)
    close(wm.grad, wr.grad, 2e-3)
    close(hbm.grad, hbr.grad, 2e-3)


def test_pack_plan_survives_data_writes_through_init_weights(hip_lib_built):
    """`.data` writes (init_weights) after a forward pass do not move version counters: the plan is invalidated explicitly."""
    import torch
    from octa_autosegmentation_amd.models.networks import DynUNet, init_weights
    torch.manual_seed(1)
    net = DynUNet(spatial_dims=2, in_channels=1, out_channels=1, kernel_size=[3, 3, 3], strides=[1, 2, 2], upsample_kernel_size=[2, 2]).cuda()
    x = torch.rand(1, 1, 32, 32, device="cuda")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        a = net(x).float()
        init_weights(net, "kaiming")                 # new weights through .data
        b = net(x).float()
        net2_out = None
    ref = DynUNet(spatial_dims=2, in_channels=1, out_channels=1, kernel_size=[3, 3, 3], strides=[1, 2, 2], upsample_kernel_size=[2, 2]).cuda()
    ref.load_state_dict(net.state_dict())
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        c = ref(x).float()
    assert not torch.equal(a, b)
    assert torch.equal(b, c)


@pytest.mark.gpu
def test_first_layer_kernels_match_torch(hip_lib_built):
    """The one-input-channel 3x3 layer (round 5 rewrite: band staged in LDS, weights / accumulators in registers, statistics epilogue,
    weight gradient through a workspace): result, statistics slots and weight gradient against torch fp32 on the same bf16-rounded
    operands. Widths that are no multiple of 8 (scalar staging), heights that are no multiple of the band, every supported Cout."""
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.models import mfma_conv as mc
    g = torch.Generator(device="cuda").manual_seed(11)
    for (n, h, w, cout) in ((2, 19, 40, 32), (1, 9, 37, 8), (3, 33, 130, 64), (1, 64, 96, 16), (2, 152, 152, 32)):
        x = torch.rand(n, h, w, 1, device="cuda", generator=g).to(torch.bfloat16)
        wt = (torch.randn(cout, 1, 3, 3, device="cuda", generator=g) / 3.0).requires_grad_(True)
        y, part = mc.conv3x3(x, wt, 1, True)
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), wt.detach(), padding=1).permute(0, 2, 3, 1)
        scale = ref.abs().max().item()
        assert y.shape == ref.shape and y.dtype == torch.bfloat16
        assert (y.float() - ref).abs().max().item() <= 2.0 ** -8 * scale + 1e-6
        # statistics: sums of the ROUNDED values per image and channel, spread over the slots
        assert part.dtype == torch.float64 and part.shape == (mc.STAT_SLOTS, n, cout, 2)
        tot = part.sum(0)
        yf = y.double()
        assert torch.allclose(tot[..., 0], yf.sum((1, 2)), rtol=1e-5, atol=1e-3 * scale)
        assert torch.allclose(tot[..., 1], (yf * yf).sum((1, 2)), rtol=1e-5, atol=1e-3 * scale * scale)
        y_plain = mc.conv3x3(x, wt, 1, False)
        assert torch.equal(y_plain, y)
        # weight gradient
        dy = torch.randn(n, h, w, cout, device="cuda", generator=g).to(torch.bfloat16)
        (gw,) = torch.autograd.grad(y, wt, dy)
        w2 = wt.detach().clone().requires_grad_(True)
        (gr,) = torch.autograd.grad(F.conv2d(x.float().permute(0, 3, 1, 2), w2, padding=1), w2, dy.float().permute(0, 3, 1, 2))
        assert (gw - gr).abs().max().item() <= 2e-4 * gr.abs().max().item() + 1e-5, (n, h, w, cout)
        # data gradient (the image is another network's output: the segmentor behind the generator): the streaming C -> 1 kernel against fp32,
        # and the weight gradient unchanged by asking for it
        xg = x.clone().requires_grad_(True)
        yg = mc.conv3x3(xg, wt, 1, False)
        assert torch.equal(yg, y)
        gx, gw2 = torch.autograd.grad(yg, (xg, wt), dy)
        x32 = x.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
        (gxr,) = torch.autograd.grad(F.conv2d(x32, wt.detach(), padding=1), x32, dy.float().permute(0, 3, 1, 2))
        gxr = gxr.permute(0, 2, 3, 1)
        assert gx.shape == x.shape and gx.dtype == torch.bfloat16
        assert (gx.float() - gxr).abs().max().item() <= 2.0 ** -8 * gxr.abs().max().item() + 1e-6, (n, h, w, cout)
        assert torch.equal(gw2, gw)
