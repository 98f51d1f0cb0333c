"""GPU: the single-precision MFMA convolution (csrc/conv_f32.hip) against torch's fp32 convolution on the CPU -- the reference's
test.py / validate.py path (fp32, no autocast) -- layer by layer and as a whole DynUNet-S at 1x1x1216x1216 (north_star: logits
within 1e-4 of the reference's fp32 CPU path)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_conv2d_f32_matches_torch_cpu(hip_lib_built):
    from octa_autosegmentation_amd.models import conv_f32
    torch.manual_seed(0)
    cases = [  # (module, input shape)
        (torch.nn.Conv2d(1, 32, 3, 1, 1, bias=False), (2, 1, 37, 45)),
        (torch.nn.Conv2d(32, 64, 3, 2, 1, bias=False), (1, 32, 64, 96)),
        (torch.nn.Conv2d(40, 72, 3, 1, 1, bias=True), (2, 40, 19, 33)),          # channel counts that are no multiples of 8 / 32
        (torch.nn.Conv2d(64, 64, 3, 2, 1, bias=False), (1, 64, 31, 29)),        # odd sizes, stride 2
        (torch.nn.Conv2d(32, 1, 1, 1, 0, bias=True), (2, 32, 50, 70)),           # the output head
        (torch.nn.Conv2d(128, 256, 3, 1, 1, bias=False), (1, 128, 40, 40)),
        (torch.nn.ConvTranspose2d(64, 32, 2, 2, bias=False), (2, 64, 21, 17)),
        (torch.nn.ConvTranspose2d(512, 256, 1, 1, bias=False), (1, 512, 24, 24)),
        (torch.nn.ConvTranspose2d(24, 20, 2, 2, bias=False), (3, 24, 9, 33)),       # ragged channel block of the one-launch 2x2 kernel
        (torch.nn.Conv2d(256, 256, 3, 1, 1, bias=False), (1, 256, 152, 152)),       # the narrow (32-channel) variant chosen for CU balance
    ]
    for mod, shape in cases:
        x = torch.randn(*shape)
        with torch.no_grad():
            ref = mod(x)
            assert conv_f32.supported(mod)
            g = mod.cuda()
            assert conv_f32.applies(g, x.cuda())
            got = conv_f32.forward(g, x.cuda()).cpu()
        assert got.shape == ref.shape and got.dtype == torch.float32
        scale = ref.abs().max().item()
        assert (got - ref).abs().max().item() <= 2e-6 * scale + 1e-6, (mod, (got - ref).abs().max().item(), scale)
    # the one-launch 2x2 transposed convolution gives the bits of four 1x1 launches with scattered stores
    mod = torch.nn.ConvTranspose2d(64, 32, 2, 2, bias=False).cuda()
    x = torch.randn(2, 64, 21, 17, device="cuda")
    with torch.no_grad():
        one = conv_f32.forward(mod, x)
        wp = conv_f32._packed(mod.weight, True)
        four = torch.full_like(one, float("nan"))
        for a in range(2):
            for b in range(2):
                conv_f32._launch(x, wp, (a * 2 + b) * 32, None, four, 32, 4 * 32, 1, 1, 0, 21, 17, 2, a, b)
    assert torch.equal(one, four)
    # a pass that records gradients stays on the torch modules
    m = torch.nn.Conv2d(8, 8, 3, 1, 1).cuda()
    assert not conv_f32.applies(m, torch.randn(1, 8, 8, 8, device="cuda"))
    with torch.no_grad():
        assert conv_f32.applies(m, torch.randn(1, 8, 8, 8, device="cuda"))
        with torch.autocast("cuda", dtype=torch.bfloat16):
            assert not conv_f32.applies(m, torch.randn(1, 8, 8, 8, device="cuda"))


def test_dynunet_fp32_logits_at_full_size_within_1e4_of_cpu(hip_lib_built, monkeypatch):
    """north_star: segmentation logits within 1e-4 (fp32) of the reference's CPU path, at the size the configs use
    (1 x 1 x 1216 x 1216), on THIS repository's kernels: every convolution on csrc/conv_f32.hip, every InstanceNorm + LeakyReLU on
    csrc/norm.hip; the vendor library's convolution is switched off for the product pass (torch.backends.cudnn.enabled = False) so
    that a silent fall-back would be slow, and the routing is asserted by counting the calls."""
    from octa_autosegmentation_amd.models import conv_f32, networks
    torch.manual_seed(0)
    net = networks.DynUNet()
    networks.init_weights(net, init_type="kaiming", nonlinearity="leaky_relu")
    x = torch.rand(1, 1, 1216, 1216)
    with torch.no_grad():
        ref = net(x)                                  # CPU fp32 torch modules
    calls = []
    orig = conv_f32.forward
    monkeypatch.setattr(conv_f32, "forward", lambda c, t: (calls.append(type(c).__name__), orig(c, t))[1])
    net = net.cuda().eval()
    old = torch.backends.cudnn.enabled
    torch.backends.cudnn.enabled = False
    try:
        with torch.no_grad():
            got = net(x.cuda()).cpu()
    finally:
        torch.backends.cudnn.enabled = old
    n_convs = sum(1 for m in net.modules() if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)))
    assert len(calls) == n_convs >= 19, (len(calls), n_convs)       # every convolution of the network went through csrc/conv_f32.hip
    err = (got - ref).abs().max().item()
    print(f"[fp32 DynUNet @1216^2] max |logit difference| {err:.2e} (logit scale {ref.abs().max().item():.2f})", flush=True)
    assert torch.allclose(got, ref, atol=1e-4, rtol=1e-4), err

