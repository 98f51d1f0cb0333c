"""The GAN networks of models/networks.py (ResNet-9 generator, 70x70 PatchGAN, blur down/up-sampling) against outputs
of the REFERENCE's models/networks.py recorded in tests/golden/networks_golden.npz (tools/make_golden_networks.py,
SURVEY.md 8c G5): same state_dict keys and shapes, and, with the same closed-form weights and inputs, the same outputs."""
import os

import numpy as np
import pytest
import torch

from octa_autosegmentation_amd.models import networks

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "networks_golden.npz"))


def fill(shape, k):        # tools/make_golden_networks.py: fill
    n = int(np.prod(shape))
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    v = np.sin(np.arange(n, dtype=np.float64) * (0.37 + 0.011 * k) + 0.5 * k) / np.sqrt(max(fan_in, 1))
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def image(shape, k):       # tools/make_golden_networks.py: image
    n = int(np.prod(shape))
    v = 0.5 + 0.5 * np.sin(np.arange(n, dtype=np.float64) * 0.0137 * (k + 1) + np.arange(n, dtype=np.float64) ** 2 * 1e-7)
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def build(name):
    net = {"G": networks.resnetGenerator9, "D": networks.patchGAN70x70}[name]().eval()
    sd = net.state_dict()
    assert list(sd.keys()) == list(G[f"{name}_keys"]), "state_dict keys differ from the reference's"
    assert [",".join(map(str, t.shape)) for t in sd.values()] == list(G[f"{name}_shapes"])
    for k, (key, t) in enumerate(sd.items()):
        if not key.endswith("filt"):
            sd[key] = fill(tuple(t.shape), k)
    net.load_state_dict(sd)
    return net


def check(name, net, size, device, tol):
    with torch.no_grad():
        y = net.to(device)(image((1, 1, size, size), 1 if name == "G" else 2).to(device)).double().cpu().numpy()
    if size == 64:
        want = G[f"{name}_out_64"]
        assert y.shape == want.shape
        assert np.abs(y - want).max() <= tol
    else:
        assert np.abs(y[0, 0, :24, :24] - G[f"{name}_crop_304"]).max() <= tol
        assert np.abs(y[0, 0, -24:, -24:] - G[f"{name}_crop2_304"]).max() <= tol
    s = G[f"{name}_sum_{size}"]
    assert abs(y.sum() - s[0]) <= tol * y.size and abs(np.abs(y).sum() - s[1]) <= tol * y.size


@pytest.mark.parametrize("name", ["G", "D"])
def test_cpu_networks_match_reference_outputs(name):
    net = build(name)
    check(name, net, 64, "cpu", 2e-5)


def test_cpu_generator_matches_reference_at_304():
    check("G", build("G"), 304, "cpu", 2e-5)


@pytest.mark.parametrize("shape", [(1, 4, 16, 16), (1, 4, 15, 13)])
def test_cpu_blur_layers_match_reference(shape):
    x, tag = image(shape, 3), f"{shape[2]}x{shape[3]}"
    assert np.abs(networks.Downsample(4)(x).numpy() - G[f"down_{tag}"]).max() <= 1e-6
    assert np.abs(networks.Upsample(4)(x).numpy() - G[f"up_{tag}"]).max() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 4, 16, 16), (1, 4, 15, 13)])
@pytest.mark.parametrize("layout", ["nchw", "nhwc"])
def test_hip_blur_kernels_match_reference(shape, layout):
    from octa_autosegmentation_amd.models import resample
    x, tag = image(shape, 3).cuda(), f"{shape[2]}x{shape[3]}"
    xin = x.permute(0, 2, 3, 1).contiguous() if layout == "nhwc" else x
    back = (lambda t: t.permute(0, 3, 1, 2)) if layout == "nhwc" else (lambda t: t)
    assert np.abs(back(resample.blur_down(xin, layout)).cpu().numpy() - G[f"down_{tag}"]).max() <= 1e-6
    assert np.abs(back(resample.blur_up(xin, layout)).cpu().numpy() - G[f"up_{tag}"]).max() <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name,size", [("G", 64), ("G", 304), ("D", 64), ("D", 304)])
def test_gpu_fp32_networks_match_reference_outputs(name, size):
    """fp32 on the GPU: pad / blur layers on the HIP kernels (exact to 1e-6 on their own, test above), the 7x7 / 4x4 / 3x3
    convolutions in MIOpen's fp32 kernels, whose summation order differs from the CPU's: through ~25 layers with instance
    norms the measured difference is 2e-4 .. 7e-4 on outputs of order 1 (a wrong tap or pad would be off by 1e-1)."""
    with networks.vendor_reference():          # fp32 GAN networks on the GPU are the torch modules (MIOpen) by design: the reference side
        check(name, build(name), size, "cuda", 2e-3)


def _mfma_out(net, x):
    before = networks.PATH_COUNTS["mfma"]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(x)
    assert networks.PATH_COUNTS["mfma"] == before + 1, "the bf16 pass did not take the hand-written kernels"
    return y.double().cpu().numpy()


# bf16 budget of the benched path against the REFERENCE's fp32 outputs (tools/make_golden_networks.py): relative L2 error of the whole
# output and the largest single deviation relative to the output's range. Measured on MI355X (round 5): G 64^2 0.4 % / 1.3 %, G 304^2
# 0.5 % / crops 1.6 %, D 64^2 0.7 % / 2.0 %, D 304^2 0.6 %; a single wrong tap in ONE of the generator's 18 residual convolutions
# gives 9 % / 30 % (test below), so the budgets sit a factor of five above the measured error and a factor of four below one wrong tap.
MFMA_REL_L2, MFMA_MAX_OVER_RANGE = 0.02, 0.08


@pytest.mark.gpu
@pytest.mark.parametrize("name,size", [("G", 64), ("G", 304), ("D", 64), ("D", 304)])
def test_gpu_mfma_networks_match_reference_outputs(name, size):
    """The reference-made outputs against the bf16 / MFMA path itself (round 4 ran them through MIOpen fp32 only): the 3x3 / 4x4 stages
    on csrc/conv.hip, stems / heads on csrc/thin_conv.hip, norms on csrc/norm.hip, pad / blur on csrc/blur.hip, under bf16 autocast."""
    net = build(name).cuda()
    y = _mfma_out(net, image((1, 1, size, size), 1 if name == "G" else 2).cuda())
    if size == 64:
        want = G[f"{name}_out_64"]
        rng = want.max() - want.min()
        rel = np.linalg.norm(y - want) / np.linalg.norm(want)
        worst = np.abs(y - want).max() / rng
        print(f"[mfma golden] {name} {size}: rel L2 {rel:.4f}, max/range {worst:.4f}")
        assert rel <= MFMA_REL_L2 and worst <= MFMA_MAX_OVER_RANGE, (rel, worst)
    else:
        for crop, key in ((y[0, 0, :24, :24], f"{name}_crop_304"), (y[0, 0, -24:, -24:], f"{name}_crop2_304")):
            want = G[key]
            rng = max(want.max() - want.min(), 1e-6)
            rel = np.linalg.norm(crop - want) / np.linalg.norm(want)
            worst = np.abs(crop - want).max() / rng
            print(f"[mfma golden] {name} {size} {key}: rel L2 {rel:.4f}, max/range {worst:.4f}")
            assert rel <= MFMA_REL_L2 and worst <= MFMA_MAX_OVER_RANGE, (key, rel, worst)
        s = G[f"{name}_sum_{size}"]
        assert abs(y.sum() - s[0]) <= 0.01 * s[1] and abs(np.abs(y).sum() - s[1]) <= 0.01 * s[1]


@pytest.mark.gpu
def test_a_wrong_tap_in_one_residual_block_breaks_the_budget():
    """The network-level budget has teeth: taps (0, 0) and (2, 2) of ONE residual convolution (block 5 of 9, second convolution)
    exchanged -- everything else untouched -- must fail the same comparison the test above passes."""
    net = build("G").cuda()
    conv = net.model[12 + 4].conv_block[5]
    with torch.no_grad():
        w = conv.weight.clone()
        conv.weight[:, :, 0, 0], conv.weight[:, :, 2, 2] = w[:, :, 2, 2], w[:, :, 0, 0]
    from octa_autosegmentation_amd.models import mfma_conv
    mfma_conv.invalidate_all_pack_plans(net)
    y = _mfma_out(net, image((1, 1, 64, 64), 1).cuda())
    want = G["G_out_64"]
    rel = np.linalg.norm(y - want) / np.linalg.norm(want)
    worst = np.abs(y - want).max() / (want.max() - want.min())
    print(f"[mfma golden] wrong tap: rel L2 {rel:.4f}, max/range {worst:.4f}")
    assert rel > MFMA_REL_L2 or worst > MFMA_MAX_OVER_RANGE, (rel, worst)
