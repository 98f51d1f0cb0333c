"""GPU parity AT THE SIZES OF BASELINE.json's configs (round 1 had these only in tools/): every DynUNet-S layer shape at B = 4
(up to 512 channels at 152x152 and 32 channels at 1216x1216: the 16-row-tile and 64-bit-offset paths of csrc/conv.hip), the 1x1
split-K weight gradient and the fused norm + head at 1216x1216, the whole network at 1216x1216 against an fp32 run on the same
bf16-rounded weights, a 128-sample full-length simulator batch against the oracle, and one 1216x1216x16 voxel volume."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (cin, cout, H, W of the INPUT, stride) of the fourteen 3x3 convolutions of DynUNet-S at 1x1216x1216 (configs/config_ves_seg-S.yml:6-13)
LAYERS = [(1, 32, 1216, 1), (32, 32, 1216, 1), (32, 64, 1216, 2), (64, 64, 608, 1), (64, 128, 608, 2), (128, 128, 304, 1), (128, 256, 304, 2),
          (256, 256, 152, 1), (256, 512, 152, 1), (512, 512, 152, 1), (512, 256, 152, 1), (256, 128, 304, 1), (128, 64, 608, 1), (64, 32, 1216, 1)]


def _tol(got, want, rel):
    err = (got.float() - want.float()).abs().max().item()
    scale = want.float().abs().max().item() + 1e-6
    assert err <= scale * rel + 1e-6, (err, scale)


@pytest.mark.parametrize("cin,cout,hw,stride", LAYERS)
def test_every_dynunet_layer_shape_at_batch_4(hip_lib_built, cin, cout, hw, stride):
    """Forward, data gradient and weight gradient of the autograd binding (models/mfma_conv.conv3x3: MFMA kernels, the one-channel
    first layer on its streaming kernels) against torch's fp32 convolution on the same bf16-rounded operands. Forward / data
    gradient: one bf16 rounding (2^-8 of the tensor scale); weight gradient: fp32 accumulation order only (1e-3)."""
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.models import mfma_conv
    B = 4
    g = torch.Generator(device="cuda").manual_seed(cin * 7 + cout + hw)
    x = torch.randn(B, hw, hw, cin, device="cuda", generator=g).to(torch.bfloat16)
    w = (torch.randn(cout, cin, 3, 3, device="cuda", generator=g) / (3.0 * cin ** 0.5)).to(torch.bfloat16).float()
    xr, wr = x.float().permute(0, 3, 1, 2).requires_grad_(True), w.clone().requires_grad_(True)
    # reference = torch's native im2col + GEMM convolution (MIOpen off: on a fresh box it would JIT-compile three kernels per shape)
    with torch.backends.cudnn.flags(enabled=False):
        yr = F.conv2d(xr, wr, stride=stride, padding=1)
        dy = torch.randn(yr.shape, device="cuda", generator=g).to(torch.bfloat16)
        yr.backward(dy.float())
    xm, wm = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ym = mfma_conv.conv3x3(xm, wm, stride)
    assert tuple(ym.shape) == (B, hw // stride, hw // stride, cout)
    ym.backward(dy.permute(0, 2, 3, 1).contiguous())
    _tol(ym.detach(), yr.detach().permute(0, 2, 3, 1), 2.0 ** -8)
    if cin > 1:                       # the network input needs no gradient
        _tol(xm.grad, xr.grad.permute(0, 2, 3, 1), 2.0 ** -8)
    _tol(wm.grad, wr.grad, 1e-3)


def test_bottleneck_1x1_transposed_conv_and_2x2_upsampling_at_full_size(hip_lib_built):
    """upsamples.0 (ConvTranspose2d 512 -> 256, k = s = 1 at 152x152: the hand-split batched weight gradient) and upsamples.3
    (k = s = 2, 64 -> 32 channels, 608 -> 1216) at B = 4."""
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.models import mfma_conv
    g = torch.Generator(device="cuda").manual_seed(5)
    for cin, cout, hw, k in ((512, 256, 152, 1), (64, 32, 608, 2)):
        x = torch.randn(4, hw, hw, cin, device="cuda", generator=g).to(torch.bfloat16)
        wt = (torch.randn(cin, cout, k, k, device="cuda", generator=g) / (k * cin ** 0.5)).to(torch.bfloat16).float()
        xr, wr = x.float().requires_grad_(True), wt.clone().requires_grad_(True)
        with torch.backends.cudnn.flags(enabled=False):
            yr = F.conv_transpose2d(xr.permute(0, 3, 1, 2), wr, stride=k).permute(0, 2, 3, 1)
            dy = torch.randn(yr.shape, device="cuda", generator=g).to(torch.bfloat16)
            yr.backward(dy.float())
        xm, wm = x.clone().requires_grad_(True), wt.clone().requires_grad_(True)
        ym = mfma_conv.conv_transpose_kxk_nhwc(xm, wm, k)
        ym.backward(dy)
        _tol(ym.detach(), yr.detach(), 2.0 ** -8)
        _tol(xm.grad, xr.grad, 2.0 ** -8)
        _tol(wm.grad, wr.grad, 1e-3)


def test_fused_norm_head_at_1216(hip_lib_built):
    """The last InstanceNorm + LeakyReLU + 1x1 output convolution as one layer (csrc/norm.hip) on a 4 x 1216 x 1216 x 32 tensor
    against the float64 torch formulation."""
    import torch
    import torch.nn.functional as F
    from octa_autosegmentation_amd.models import mfma_conv
    g = torch.Generator(device="cuda").manual_seed(9)
    n, h, w, c = 4, 1216, 1216, 32
    x = (torch.randn(n, h, w, c, device="cuda", generator=g) * 1.7 + 0.3).to(torch.bfloat16)
    gamma, beta = torch.rand(c, device="cuda", generator=g) + 0.5, torch.randn(c, device="cuda", generator=g) * 0.2
    hw_, hb = torch.randn(1, c, 1, 1, device="cuda", generator=g) / c ** 0.5, torch.randn(1, device="cuda", generator=g)
    dl = torch.randn(n, h, w, 1, device="cuda", generator=g).to(torch.bfloat16)
    xr, gr, br, wr, hbr = (t.clone().double().requires_grad_(True) for t in (x, gamma, beta, hw_, hb))
    y = F.leaky_relu(F.instance_norm(xr.permute(0, 3, 1, 2), weight=gr, bias=br, eps=1e-5), 0.01)
    lr = (y.permute(0, 2, 3, 1) @ wr.reshape(-1, 1) + hbr)
    lr.backward(dl.double())
    xm, gm, bm, wm, hbm = (t.clone().requires_grad_(True) for t in (x, gamma, beta, hw_, hb))
    lm = mfma_conv.instance_norm_leaky_relu_head1_nhwc(xm, gm, bm, 0.01, 1e-5, wm, hbm)
    lm.backward(dl)
    _tol(lm.detach(), lr.detach(), 2.0 ** -8)
    _tol(xm.grad, xr.grad, 2.0 ** -7)
    for a, b in ((gm.grad, gr.grad), (bm.grad, br.grad), (wm.grad, wr.grad), (hbm.grad, hbr.grad)):
        _tol(a, b, 2e-3)


def test_whole_network_at_1216_against_fp32_on_the_same_bf16_weights(hip_lib_built):
    """DynUNet-S at 2 x 1 x 1216 x 1216: logits and every parameter gradient of the MFMA path against the torch fp32 modules
    carrying the SAME bf16-rounded parameters, as per-tensor relative error (||got - ref|| / ||ref||), next to torch's own bf16
    autocast on the same problem as the yardstick of what bf16 activations cost. north_star's "logits within 1e-4 fp32" is a
    statement about the fp32 path (tests/test_models_gpu.py::test_dynunet_logits_match_cpu_fp32); the benched path computes in
    bf16 and is held to bf16's error budget here."""
    import torch
    from octa_autosegmentation_amd.models import networks
    torch.manual_seed(3)
    net = networks.DynUNet(2, 1, 1, [3, 3, 3, 3, 3], [1, 2, 2, 2, 1], [1, 2, 2, 2, 1]).cuda()
    networks.init_weights(net, "kaiming")
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    x = torch.rand(2, 1, 1216, 1216, device="cuda").to(torch.bfloat16).float()
    tgt = (torch.rand(2, 1, 1216, 1216, device="cuda") > 0.7).float()

    def run(mfma, autocast):
        old = (networks.USE_MFMA_CONV, networks.USE_FUSED_NORM)
        networks.USE_MFMA_CONV, networks.USE_FUSED_NORM = mfma, mfma
        try:
            net.zero_grad(set_to_none=True)
            # the torch reference modules run torch's native kernels (MIOpen off: on a fresh box its JIT compilation of every fp32 and
            # bf16 layer shape takes four minutes); the MFMA path does not touch MIOpen either way
            with torch.backends.cudnn.flags(enabled=False), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
                out = net(x)
                torch.nn.functional.binary_cross_entropy_with_logits(out.float(), tgt).backward()
            return out.float().detach(), {k: p.grad.float().clone() for k, p in net.named_parameters() if p.grad is not None}
        finally:
            networks.USE_MFMA_CONV, networks.USE_FUSED_NORM = old

    ref_out, ref_g = run(False, False)
    ac_out, ac_g = run(False, True)
    got_out, got_g = run(True, True)
    rel = lambda a, b: ((a - b).norm() / (b.norm() + 1e-30)).item()
    e_got, e_ac = rel(got_out, ref_out), rel(ac_out, ref_out)
    assert e_got <= 1.5 * e_ac + 0.005, (e_got, e_ac)
    worst = (0.0, "", 0.0)
    for k, b in ref_g.items():
        if b.norm().item() < 1e-12 or k not in got_g:
            continue
        r_got, r_ac = rel(got_g[k], b), rel(ac_g[k], b)
        worst = max(worst, (r_got, k, r_ac))
        assert r_got <= 2.0 * r_ac + 0.01, (k, r_got, r_ac)      # round 4 allowed a flat 10 % on top
    print(f"logits: rel. error {e_got:.4f} (torch autocast {e_ac:.4f}); worst gradient tensor {worst[1]}: {worst[0]:.4f} (torch autocast {worst[2]:.4f})")


def _oracle_one(args):
    cfg, seed = args
    from oracle import sim_oracle
    e, info = sim_oracle.simulate(cfg, seed)
    return seed, e, info["n_art_edges"]


def test_128_sample_full_length_batch_against_the_oracle(hip_lib_built):
    """BASELINE configs[1] at full size: ONE 128-sample batch of full-length runs (I = 100 + 150, N = 2000) on the GPU; eight of
    its samples (spread over the batch) are recomputed by the CPU oracle on the box's host cores: CSV bytes identical, radii
    bit-identical. Every sample of the batch must finish without error bits and with a plausible size."""
    from multiprocessing import get_context
    from octa_autosegmentation_amd import graph_io
    from octa_autosegmentation_amd.utils import configs
    from octa_autosegmentation_amd.vessel_graph_generation import greenhouse
    cfg = configs.load_generator_config()
    seeds = list(range(31000, 31127)) + [953121]      # the last one peaks at 8353 live CO2 sources (above round 1's capacity of 8192)
    picks = [0, 17, 38, 59, 64, 90, 111, 127]
    with get_context("spawn").Pool(min(8, os.cpu_count() or 1)) as pool:
        fut = pool.map_async(_oracle_one, [(cfg, seeds[k]) for k in picks])
        res = greenhouse.simulate_batch(cfg, seeds)
        ref = fut.get(timeout=900)
    assert int(res.stats[:, 0].max()) == 0
    n_edges = np.diff(res.edge_off)
    assert n_edges.min() > 9000 and n_edges.max() < 14336 * 2
    for k, (seed, e, na) in zip(picks, ref):
        gpu = res.sample_edges(k)
        assert gpu.shape == e.shape and int(res.n_art[k]) == na, seed
        assert graph_io.edges_to_csv_bytes(gpu) == graph_io.edges_to_csv_bytes(e), seed
        assert (gpu[:, 6] == e[:, 6]).all(), seed
        assert (gpu == e).all(), (seed, int((gpu != e).sum()))         # every double: radii and node positions


def test_voxel_volume_at_1216x1216x16_against_the_oracle(hip_lib_built):
    """north_star's grid: one full graph voxelised to 1216 x 1216 x 16 (padded to 53 in z like the reference, tree2img.py:206-211)
    by the HIP kernel and by the oracle: 0 differing voxels."""
    import torch
    from octa_autosegmentation_amd.utils import configs
    from octa_autosegmentation_amd.vessel_graph_generation import greenhouse, tree2img
    from oracle import octa_oracle
    cfg = configs.load_generator_config()
    cfg["Greenhouse"]["modes"][0]["I"], cfg["Greenhouse"]["modes"][1]["I"] = 60, 40
    edges = greenhouse.simulate_batch(cfg, [5]).sample_edges(0)
    assert len(edges) > 3000
    d = torch.from_numpy(np.ascontiguousarray(edges)).cuda()
    vol = tree2img.voxelize_edges_device(d, np.array([0, len(edges)]), [1216, 1216, 16])[0].cpu().numpy().view(np.uint16)
    assert vol.shape == (1216, 1216, 53)
    want = octa_oracle.voxelize(edges, [1216, 1216, 16])
    assert want.shape == vol.shape and (vol == want).all(), int((vol != want).sum())
    assert int((vol > 0).sum()) > 100000
