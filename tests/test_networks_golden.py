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


def _max_err(name, y, size):
    if size == 64:
        return float(np.abs(y - G[f"{name}_out_64"]).max())
    return float(max(np.abs(y[0, 0, :24, :24] - G[f"{name}_crop_304"]).max(), np.abs(y[0, 0, -24:, -24:] - G[f"{name}_crop2_304"]).max()))


@pytest.mark.gpu
@pytest.mark.parametrize("name,size", [("G", 64), ("G", 304), ("D", 64), ("D", 304)])
def test_gpu_fp32_networks_match_reference_outputs(name, size):
    """fp32 on the GPU WITHOUT the vendor libraries (round 6): a gradient-free fp32 pass -- what test.py / validate.py and the frozen
    generator of the data pipeline run in the reference (test.py:75-82, docker/dockershell.sh:14-16) -- takes the exact-fp32 MFMA
    convolutions (csrc/conv_f32.hip: 7x7, 4x4, 3x3, bias included), the fp32 instance norm (csrc/norm.hip) and the pad / blur kernels
    (csrc/blur.hip); not one MIOpen kernel. Against the reference-made outputs the only difference left is the summation order, which
    this fixture's nearly constant channels amplify through ~25 instance norms: the torch modules on MIOpen (the yardstick, measured in
    the same test) differ from the CPU reference by 2e-4 .. 7e-4; the own kernels must do at least as well as 1.5 x that and stay under
    1e-3 absolutely (a wrong tap or pad is off by 1e-1)."""
    x = image((1, 1, size, size), 1 if name == "G" else 2).cuda()
    net = build(name).cuda()
    before = dict(networks.PATH_COUNTS)
    with torch.no_grad():
        y = net(x).double().cpu().numpy()
    assert networks.PATH_COUNTS["f32"] == before.get("f32", 0) + 1 and networks.PATH_COUNTS["vendor"] == before.get("vendor", 0), \
        "the fp32 pass left the hand-written kernels"
    with networks.vendor_reference(), torch.no_grad():          # the same modules through torch (MIOpen): the yardstick
        y_v = net.model(x).double().cpu().numpy()
    err, err_v = _max_err(name, y, size), _max_err(name, y_v, size)
    print(f"[fp32 golden] {name} {size}: own kernels {err:.2e}, MIOpen {err_v:.2e}", flush=True)
    assert err <= max(1.5 * err_v, 1e-4) and err <= 1e-3, (err, err_v)
    s = G[f"{name}_sum_{size}"]
    assert abs(y.sum() - s[0]) <= 1e-3 * y.size and abs(np.abs(y).sum() - s[1]) <= 1e-3 * y.size


@pytest.mark.gpu
def test_gpu_fp32_generator_is_within_1e4_of_the_cpu_modules_on_conditioned_weights():
    """north_star's fp32 tolerance (1e-4) on the generator with He-initialised weights, where the summation order is not amplified: the
    own fp32 kernels against the SAME modules on the CPU, 1 x 1 x 304 x 304 (the shape test.py --General.inference G runs)."""
    torch.manual_seed(3)
    net = networks.resnetGenerator9().eval()
    networks.init_weights(net, "kaiming")
    x = image((1, 1, 304, 304), 1)
    with torch.no_grad():
        want = net(x).double().numpy()
        before = networks.PATH_COUNTS["vendor"]
        got = net.cuda()(x.cuda()).double().cpu().numpy()
    assert networks.PATH_COUNTS["vendor"] == before
    err = float(np.abs(got - want).max())
    print(f"[fp32 conditioned] generator 304^2: max |difference| {err:.2e} on outputs in [{want.min():.3f}, {want.max():.3f}]", flush=True)
    assert err <= 1e-4


def _mfma_out(net, x):
    before = networks.PATH_COUNTS["mfma"]
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        y = net(x)
    assert networks.PATH_COUNTS["mfma"] == before + 1, "the bf16 pass did not take the hand-written kernels"
    return y.double().cpu().numpy()


def _cpu_autocast_error(name, size):
    """Yardstick: the SAME torch modules on the CPU under bf16 autocast against the reference-made outputs."""
    net = build(name)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        y = net(image((1, 1, size, size), 1 if name == "G" else 2)).double().numpy()
    return y


def _errors(y, want):
    return np.linalg.norm(y - want) / np.linalg.norm(want), np.abs(y - want).max() / max(want.max() - want.min(), 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("name,size", [("G", 64), ("G", 304), ("D", 64), ("D", 304)])
def test_gpu_mfma_networks_match_reference_outputs(name, size):
    """The reference-made outputs against the bf16 / MFMA path ITSELF (round 4 ran them through MIOpen fp32 only): the 3x3 / 4x4 stages on
    csrc/conv.hip, stems / heads on csrc/thin_conv.hip, norms on csrc/norm.hip, pad / blur on csrc/blur.hip, under bf16 autocast.
    Stated bf16 budget: the fixture's closed-form weights (sin(n * 0.37 ...) / sqrt(fan_in)) produce nearly constant channels that the
    instance norms blow up again, so bf16 storage costs 3 - 12 % relative L2 on THIS fixture whoever computes it -- the same torch modules
    under bf16 autocast on the CPU measure 2.0 - 3.6 % (G) and 22 - 63 % (D). The product path must stay within 1.5 x that yardstick + 0.5 %
    (measured on MI355X: G 64^2 3.0 %, G 304^2 crops 1.8 - 3.5 %, D 64^2 5.7 %, D 304^2 crops 11.7 - 12.4 %). What this fixture cannot do under a bf16
    budget is expose a single wrong tap (0.75 % on it, below the bf16 noise): tests/test_models_gpu.py does that on He-initialised
    weights against an fp32 run (one wrong tap: 11 % against a 2.5 % budget)."""
    net = build(name).cuda()
    y = _mfma_out(net, image((1, 1, size, size), 1 if name == "G" else 2).cuda())
    yc = _cpu_autocast_error(name, size)
    if size == 64:
        pairs = [(y, yc, G[f"{name}_out_64"], "out_64")]
    else:
        pairs = [(y[0, 0, :24, :24], yc[0, 0, :24, :24], G[f"{name}_crop_304"], "crop_304"), (y[0, 0, -24:, -24:], yc[0, 0, -24:, -24:], G[f"{name}_crop2_304"], "crop2_304")]
    for got, yard, want, key in pairs:
        (rel, worst), (rel_y, worst_y) = _errors(got, want), _errors(yard, want)
        print(f"[mfma golden] {name} {size} {key}: rel L2 {rel:.4f} (cpu autocast {rel_y:.4f}), max/range {worst:.4f} ({worst_y:.4f})", flush=True)
        assert rel <= 1.5 * rel_y + 0.005 and worst <= 1.5 * worst_y + 0.02, (key, rel, rel_y, worst, worst_y)
    s = G[f"{name}_sum_{size}"]
    assert abs(y.sum() - s[0]) <= 0.05 * s[1] and abs(np.abs(y).sum() - s[1]) <= 0.05 * s[1]      # D at 304^2: 2.2 % at 12 % relative L2
