"""GPU: DynUNet forward parity (fp32, same weights) between cuda and cpu within the 1e-4 of north_star,
and one bf16 training step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_dynunet_logits_match_cpu_fp32():
    from octa_autosegmentation_amd.models import networks
    torch.manual_seed(0)
    net = networks.DynUNet()
    networks.init_weights(net, init_type="kaiming", nonlinearity="leaky_relu")
    x = torch.rand(2, 1, 96, 96)
    with torch.no_grad():
        ref = net(x)
        got = net.cuda()(x.cuda()).cpu()          # fused HIP InstanceNorm+LeakyReLU path
    assert torch.allclose(got, ref, atol=1e-4, rtol=1e-4), float((got - ref).abs().max())


def test_training_step_bf16_runs_and_learns():
    from tests.test_models import CFG
    from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
    torch.manual_seed(1)
    tr = SegmentationTrainer(CFG, "cuda")
    x = torch.rand(2, 1, 128, 128, device="cuda")
    y = (x > 0.7).float()
    first = None
    for i in range(30):
        _, losses = tr.perform_training_step({"image": x, "label": y})
        v = float(losses["DiceBCELoss"])
        assert v == v
        first = v if first is None else first
    assert v < first


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_instnorm_lrelu_matches_torch(dtype):
    from octa_autosegmentation_amd.models.fused_ops import instance_norm_leaky_relu
    torch.manual_seed(0)
    for shape in [(2, 5, 37, 41), (3, 32, 64, 64), (1, 8, 304, 304)]:
        x = (torch.randn(*shape, device="cuda") * 2 + 0.7).to(dtype).requires_grad_(True)
        w = torch.randn(shape[1], device="cuda", requires_grad=True)
        b = torch.randn(shape[1], device="cuda", requires_grad=True)
        dy = torch.randn(*shape, device="cuda").to(dtype)
        y = instance_norm_leaky_relu(x, w, b, 0.01, 1e-5)
        y.backward(dy)
        gx, gw, gb = x.grad.clone(), w.grad.clone(), b.grad.clone()
        # float64 torch reference (MIOpen's own fp32 instance-norm backward is off by up to 6e-3 on odd plane sizes)
        x2 = x.detach().double().requires_grad_(True); w2 = w.detach().double().requires_grad_(True); b2 = b.detach().double().requires_grad_(True)
        ref = torch.nn.functional.leaky_relu(torch.nn.functional.instance_norm(x2, weight=w2, bias=b2, eps=1e-5), 0.01)
        ref.backward(dy.double())
        tol = 1e-4 if dtype == torch.float32 else 2e-2
        assert torch.allclose(y.double(), ref, atol=tol, rtol=tol), (shape, float((y.double() - ref).abs().max()))
        assert torch.allclose(gx.double(), x2.grad, atol=tol, rtol=tol), (shape, float((gx.double() - x2.grad).abs().max()))
        assert torch.allclose(gw.double(), w2.grad, atol=tol * 50, rtol=tol), float((gw.double() - w2.grad).abs().max())
        assert torch.allclose(gb.double(), b2.grad, atol=tol * 50, rtol=tol)


def test_fused_dice_bce_matches_torch():
    """csrc/loss.hip against the torch composition of the same loss (fp32): value to 1e-6, gradient to 1e-6 relative."""
    from octa_autosegmentation_amd.models import losses
    g = torch.Generator(device="cuda").manual_seed(4)
    x = (torch.randn(3, 1, 157, 201, device="cuda", generator=g) * 3).requires_grad_(True)
    y = (torch.rand(3, 1, 157, 201, device="cuda", generator=g) > 0.8).float()
    loss = losses.DiceBCELoss(True)
    losses.USE_FUSED_LOSS = False
    try:
        ref = loss(x, y)
        ref.backward()
        gref = x.grad.clone()
    finally:
        losses.USE_FUSED_LOSS = True
    x.grad = None
    got = loss(x, y)
    got.backward()
    assert abs(float(got) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref)))
    assert (x.grad - gref).abs().max().item() <= 1e-6 * gref.abs().max().item() + 1e-12
    xb = x.detach().to(torch.bfloat16).requires_grad_(True)      # bf16 logits straight from the network
    gb = loss(xb, y)
    gb.backward()
    assert abs(float(gb) - float(loss(xb.detach().float(), y))) <= 1e-5 and xb.grad.dtype == torch.bfloat16


def test_gan_seg_step_on_gpu_trains_generator_through_mfma_segmentor():
    """configs[3]: the GAN-seg step under bf16 autocast; the generator's gradient flows through the segmentor's first
    (channel-padded) layer on the MFMA path."""
    from tests.test_models import CFG
    from octa_autosegmentation_amd.models.gan_seg_trainer import GanSegTrainer
    cfg = {"General": {"amp": True, "model": {"name": "GanSegModel", "model_g": {"name": "resnetGenerator9"},
                                                "model_d": {"name": "patchGAN70x70"},
                                                "model_s": dict(CFG["General"]["model"]), "upshape": (128, 128)}},
           "Train": {"lr": 2e-4, "loss_dg": "LSGANLoss", "loss_s": "DiceBCELoss"}}
    torch.manual_seed(0)
    tr = GanSegTrainer(cfg, "cuda")
    batch = {"real_A": torch.rand(2, 1, 64, 64), "real_B": torch.rand(2, 1, 64, 64), "real_A_seg": (torch.rand(2, 1, 128, 128) > 0.7).float()}
    g0 = [p.detach().clone() for p in tr.generator.parameters()][0]
    for _ in range(2):
        out, losses = tr.perform_training_step(batch)
    assert all(torch.isfinite(v) for v in losses.values())
    assert not torch.equal(g0, [p for p in tr.generator.parameters()][0])
    assert out["prediction"].shape == (1, 1, 128, 128)


def _rel(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _three_way(net, x, target):
    """One forward + backward of `net` three ways with the SAME bf16-rounded parameters: the torch modules in fp32 (the reference), the
    torch modules under bf16 autocast (the yardstick of what bf16 activations cost through this depth), and the product path (bf16
    autocast on the hand-written kernels, counted). Returns {way: (output, input gradient, {name: weight gradient})}."""
    from octa_autosegmentation_amd.models import networks
    res = {}
    for way in ("fp32", "autocast", "mfma"):
        net.zero_grad(set_to_none=True)
        xi = x.clone().requires_grad_(True)
        before = networks.PATH_COUNTS["mfma"]
        if way == "mfma":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = net(xi)
            assert networks.PATH_COUNTS["mfma"] == before + 1, "the bf16 pass did not take the hand-written kernels"
        else:
            with networks.vendor_reference(), torch.backends.cudnn.flags(enabled=False), torch.autocast("cuda", dtype=torch.bfloat16, enabled=way == "autocast"):
                old, networks.USE_MFMA_CONV = networks.USE_MFMA_CONV, False
                try:
                    y = net(xi)
                finally:
                    networks.USE_MFMA_CONV = old
            assert networks.PATH_COUNTS["mfma"] == before
        ((y.float() - target) ** 2).mean().backward()
        res[way] = (y.float().detach(), xi.grad.detach().float(), {k: p.grad.detach().float().clone() for k, p in net.named_parameters()
                                                                   if p.grad is not None and p.dim() == 4})
    return res


def _assert_within_bf16_budget(res, tag):
    """Per-tensor relative L2 error of the product path against the fp32 run: at most twice the torch-autocast run's own error + 1 %
    (output, input gradient and EVERY convolution weight gradient)."""
    ref, ac, got = res["fp32"], res["autocast"], res["mfma"]
    rows = [("output", _rel(got[0], ref[0]), _rel(ac[0], ref[0])), ("d/dx", _rel(got[1], ref[1]), _rel(ac[1], ref[1]))]
    assert set(got[2]) == set(ref[2]), "a convolution weight is missing its gradient on the product path"
    rows += [(k, _rel(got[2][k], ref[2][k]), _rel(ac[2][k], ref[2][k])) for k in ref[2] if ref[2][k].norm().item() > 1e-12]
    worst = max(rows, key=lambda r: r[1] - 2.0 * r[2])
    print(f"[{tag}] output {rows[0][1]:.4f} (torch autocast {rows[0][2]:.4f}), d/dx {rows[1][1]:.4f} ({rows[1][2]:.4f}), worst tensor {worst[0]}: "
          f"{worst[1]:.4f} ({worst[2]:.4f})", flush=True)
    for name, r_got, r_ac in rows:
        assert r_got <= 2.0 * r_ac + 0.01, (tag, name, r_got, r_ac)
    return rows


def _bf16_params(net):
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(p.to(torch.bfloat16).float())
    from octa_autosegmentation_amd.models import mfma_conv
    mfma_conv.invalidate_all_pack_plans(net)


def test_resnet_generator_mfma_path_against_fp32_on_the_same_weights():
    """ResNet-9 generator (reference models/networks.py:291-443), He-initialised: output, input gradient and every convolution weight
    gradient of the bf16 / MFMA path against an fp32 run of the torch modules on the same bf16-rounded weights (round 4 compared with
    torch's own autocast by cosine > 0.9, which a wrong tap in one of nine blocks would have passed)."""
    from octa_autosegmentation_amd.models import networks
    torch.manual_seed(2)
    g = networks.resnetGenerator9().cuda()
    networks.init_weights(g, "kaiming", nonlinearity="relu")
    _bf16_params(g)
    x = torch.rand(2, 1, 128, 128, device="cuda").to(torch.bfloat16).float()
    target = torch.rand(2, 1, 128, 128, device="cuda")
    _assert_within_bf16_budget(_three_way(g, x, target), "generator")


def test_a_wrong_tap_in_one_residual_block_fails_the_budget():
    """Teeth of the test above: the product path runs with taps (0, 0) and (2, 2) of ONE residual convolution exchanged (block 5 of 9,
    second convolution; the fp32 reference keeps the right weights) -- the comparison must fail."""
    from octa_autosegmentation_amd.models import mfma_conv, networks
    torch.manual_seed(2)
    g = networks.resnetGenerator9().cuda()
    networks.init_weights(g, "kaiming", nonlinearity="relu")
    _bf16_params(g)
    x = torch.rand(2, 1, 128, 128, device="cuda").to(torch.bfloat16).float()
    target = torch.rand(2, 1, 128, 128, device="cuda")
    conv = g.model[12 + 4].conv_block[5]
    real_pack = mfma_conv.pack_weight

    def wrong_pack(w, cin_pad=None):
        t = real_pack(w, cin_pad)
        if w is conv.weight:
            t = t.clone()
            t[0], t[8] = t[8].clone(), t[0].clone()
        return t
    mfma_conv.pack_weight = wrong_pack
    try:
        res = _three_way(g, x, target)
    finally:
        mfma_conv.pack_weight = real_pack
    r_got, r_ac = _rel(res["mfma"][0], res["fp32"][0]), _rel(res["autocast"][0], res["fp32"][0])
    print(f"[generator, one wrong tap] output {r_got:.4f} (torch autocast {r_ac:.4f}; budget {2 * r_ac + 0.01:.4f})", flush=True)
    assert r_got > 2.0 * r_ac + 0.01, (r_got, r_ac)
    with pytest.raises(AssertionError):
        _assert_within_bf16_budget(res, "generator, one wrong tap")


def test_conv4x4_mfma_matches_torch():
    """4x4 stride-1 padding-1 convolution (PatchGAN inner layers) on the MFMA kernel: forward, data gradient, weight gradient
    against torch fp32 on the same bf16-rounded operands, odd sizes included."""
    from octa_autosegmentation_amd.models import mfma_conv as mc
    torch.manual_seed(11)
    for (n, h, w, cin, cout) in [(2, 19, 37, 32, 64), (1, 75, 75, 128, 256), (2, 8, 8, 64, 32), (1, 151, 151, 64, 128)]:
        x = torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16)
        wgt = (torch.randn(cout, cin, 4, 4, device="cuda") / (4 * cin ** 0.5)).to(torch.bfloat16).float()
        xa = x.clone().requires_grad_(True)
        wa = wgt.clone().requires_grad_(True)
        y = mc.conv4x4(xa, wa)
        xr = x.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
        wr = wgt.clone().requires_grad_(True)
        yr = torch.nn.functional.conv2d(xr, wr, padding=1)
        assert y.shape == (n, h - 1, w - 1, cout)
        dy = torch.randn_like(yr).to(torch.bfloat16)
        y.backward(dy.permute(0, 2, 3, 1).contiguous())
        yr.backward(dy.float())
        sc = lambda t: max(t.abs().max().item(), 1e-6)
        assert (y.float().permute(0, 3, 1, 2) - yr).abs().max().item() <= 2 ** -7 * sc(yr)
        assert (xa.grad.float().permute(0, 3, 1, 2) - xr.grad).abs().max().item() <= 2 ** -7 * sc(xr.grad)
        assert (wa.grad - wr.grad).abs().max().item() <= 1e-3 * sc(wr.grad)       # fp32 accumulation order only


def test_patchgan_mfma_path_against_fp32_on_the_same_weights():
    """PatchGAN (reference models/networks.py:445-506), He-initialised: output, input gradient (what trains the generator) and every
    convolution weight gradient of the bf16 / MFMA path against an fp32 run of the torch modules on the same bf16-rounded weights."""
    from octa_autosegmentation_amd.models import networks
    torch.manual_seed(4)
    net = networks.patchGAN70x70().cuda()
    networks.init_weights(net, "kaiming")
    _bf16_params(net)
    x = torch.rand(2, 1, 128, 128, device="cuda").to(torch.bfloat16).float()
    res = _three_way(net, x, torch.zeros(2, 1, 14, 14, device="cuda"))
    assert res["mfma"][0].shape == (2, 1, 14, 14)
    _assert_within_bf16_budget(res, "discriminator")


# ---- a21 on the device (round 3): the reference-made fixture of the joint G / D / S update on cuda --------------------------------

def _report(tag, what, got, want):
    rel = np.abs(np.asarray(got) - np.asarray(want)) / np.maximum(np.abs(np.asarray(want)), 1e-12)
    print(f"[ganseg {tag}] {what}: max rel dev {rel.max():.3e}", flush=True)
    return rel



@pytest.mark.parametrize("tag,idt", [("he_idt0", False), ("he_idt1", True)])
def test_gan_seg_fixture_on_cuda_fp32(tag, idt):
    """tests/golden/ganseg_golden.npz (two perform_training_steps of the reference's own GanSegModel, tools/make_golden_ganseg.py;
    reference models/gan_seg_model.py:116-173; the well-conditioned `he_` parameters) with `device: cuda, amp: False`: the HIP pad /
    blur / InstanceNorm kernels and fp32 convolutions. Tolerances (measured on MI355X: 1.4e-5 / 3.2e-4 / 1e-4 / 6.7e-3, and on another box -- the
    vendor library picks its fp32 algorithms per run -- 3.3e-6 / 1.3e-4 / 1.2e-3 / 2.4e-2): step-1 losses 1e-4 relative, step-1 gradient
    norms 1e-3 (one backward pass, no update behind it), step-2 losses 5e-3 and step-2 gradient norms 5e-2 (the first Adam step is
    lr * sign(g): parameters with rounding-noise gradients move either way)."""
    from octa_autosegmentation_amd.models import networks
    from tests.test_models import run_gan_seg_fixture
    with networks.vendor_reference():         # `amp: False` on the GPU is the torch modules' fp32 path by design (loud elsewhere: OCTA_STRICT)
        losses, gnorm, sums, g = run_gan_seg_fixture(tag, idt, device="cuda", amp=False)
    _report(tag, "fp32 step-1 losses", losses[0], g[f"{tag}_losses"][0])
    _report(tag, "fp32 step-2 losses", losses[1], g[f"{tag}_losses"][1])
    _report(tag, "fp32 step-1 grad norms", gnorm[0], g[f"{tag}_grad_norms_steps"][0])
    _report(tag, "fp32 step-2 grad norms", gnorm[1], g[f"{tag}_grad_norms_steps"][1])
    assert np.allclose(losses[0], g[f"{tag}_losses"][0], rtol=1e-4, atol=1e-6), (losses[0], g[f"{tag}_losses"][0])
    assert np.allclose(gnorm[0], g[f"{tag}_grad_norms_steps"][0], rtol=1e-3), (gnorm[0], g[f"{tag}_grad_norms_steps"][0])
    assert np.allclose(losses[1], g[f"{tag}_losses"][1], rtol=5e-3, atol=1e-6), (losses[1], g[f"{tag}_losses"][1])
    assert np.allclose(gnorm[1], g[f"{tag}_grad_norms_steps"][1], rtol=5e-2), (gnorm[1], g[f"{tag}_grad_norms_steps"][1])
    assert np.allclose(sums[:, 1], g[f"{tag}_param_sums"][:, 1], rtol=1e-5), (sums, g[f"{tag}_param_sums"])


@pytest.mark.parametrize("tag,idt", [("he_idt0", False), ("he_idt1", True)])
def test_gan_seg_fixture_on_cuda_bf16_mfma(tag, idt):
    """The same fixture through the PRODUCT path of the GAN-seg step: `amp: True` -- bf16 autocast, generator / PatchGAN / DynUNet
    on the MFMA convolution, thin-conv, NHWC InstanceNorm and fused loss kernels, passes batched over concatenated mini-batches.
    bf16 budget (8 mantissa bits, ~25 layers deep, 32x32 inputs; measured on MI355X: step-1 losses 0.9 %, step-1 gradient norms
    0.1 - 4.3 %, step-2 losses 12 % with one tap order of the convolution kernels and 22 % with another): step-1 losses within 3 %,
    step-1 gradient norms of G / D / S within 8 % -- the pin of the
    backward composition: a missing detach of fake_B in the D pass adds D's gradient to G's (norm x1.4), a D that is not frozen in
    the G+S pass doubles D's, attached pseudo-labels change S's by tens of percent; step-2 quantities within 35 %: they sit behind
    the first Adam step, lr * sign(g) for every parameter, so a parameter whose gradient is rounding noise moves either way and the
    discriminator's second loss follows the fp32 summation order inside the convolutions -- a sanity bound, not a pin (the fp32
    test above is the pin of step 2)."""
    from tests.test_models import run_gan_seg_fixture
    losses, gnorm, sums, g = run_gan_seg_fixture(tag, idt, device="cuda", amp=True)
    _report(tag, "bf16 step-1 losses", losses[0], g[f"{tag}_losses"][0])
    _report(tag, "bf16 step-2 losses", losses[1], g[f"{tag}_losses"][1])
    _report(tag, "bf16 step-1 grad norms", gnorm[0], g[f"{tag}_grad_norms_steps"][0])
    _report(tag, "bf16 step-2 grad norms", gnorm[1], g[f"{tag}_grad_norms_steps"][1])
    assert np.allclose(losses[0], g[f"{tag}_losses"][0], rtol=3e-2, atol=1e-3), (losses[0], g[f"{tag}_losses"][0])
    assert np.allclose(gnorm[0], g[f"{tag}_grad_norms_steps"][0], rtol=8e-2), (gnorm[0], g[f"{tag}_grad_norms_steps"][0])
    assert np.allclose(losses[1], g[f"{tag}_losses"][1], rtol=0.35, atol=1e-3), (losses[1], g[f"{tag}_losses"][1])
    assert np.allclose(gnorm[1], g[f"{tag}_grad_norms_steps"][1], rtol=0.35), (gnorm[1], g[f"{tag}_grad_norms_steps"][1])
    assert np.allclose(sums[:, 1], g[f"{tag}_param_sums"][:, 1], rtol=1e-4), (sums, g[f"{tag}_param_sums"])


@pytest.fixture(scope="module")
def one_rank_rccl():
    """A one-rank `nccl` (= RCCL) process group in this process: device-memory all-reduce and communicator set-up are real."""
    import os
    import socket
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_grad_arena_and_rccl_on_device_equal_the_plain_step(one_rank_rccl):
    """(e) on the GPU: OCTA_GRAD_ARENA=1 makes every parameter's .grad a view into one flat fp32 buffer per optimiser, the MFMA
    autograd functions accumulate into the views, the exchange is ONE RCCL all-reduce on that buffer (one rank: sum = identity).
    Both trainers give the arena-less step's numbers: the GAN-seg fixture (three optimisers, bf16 / MFMA) and the U-Net step."""
    from tests.test_models import CFG, run_gan_seg_fixture
    plain = run_gan_seg_fixture("he_idt1", True, device="cuda", amp=True)
    arena = run_gan_seg_fixture("he_idt1", True, device="cuda", amp=True, arena=True)
    # Step 1 (same weights, same inputs): the forward pass is deterministic, the gradients are reproducible to the order of the fp32
    # atomics in the weight-gradient kernels. Step 2 sits behind the first Adam step, lr * sign(g): a parameter whose gradient is at
    # rounding level moves either way, so two runs of the SAME configuration differ there (measured run to run, no arena involved:
    # up to 0.6 % in the discriminator's loss) -- a loose bound only.
    assert np.allclose(plain[0][0], arena[0][0], rtol=1e-5, atol=1e-6), (plain[0], arena[0])
    assert np.allclose(plain[1][0], arena[1][0], rtol=2e-3), (plain[1], arena[1])
    assert np.allclose(plain[0][1], arena[0][1], rtol=5e-2, atol=1e-4), (plain[0], arena[0])
    assert np.allclose(plain[1][1], arena[1][1], rtol=0.1), (plain[1], arena[1])
    import os
    from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
    x = torch.rand(2, 1, 128, 128, device="cuda")
    y = (x > 0.7).float()
    traj = {}
    for use in (False, True):
        if use:
            os.environ["OCTA_GRAD_ARENA"] = "1"
        try:
            torch.manual_seed(1)
            tr = SegmentationTrainer(CFG, "cuda")
        finally:
            os.environ.pop("OCTA_GRAD_ARENA", None)
        assert bool(tr.impl._arenas) == use
        vals = []
        for _ in range(6):
            _, l = tr.perform_training_step({"image": x, "label": y})
            vals.append(float(l["DiceBCELoss"]))
        if use:
            a = tr.impl._arenas["optimizer"]
            a.check_views()
            assert a.flat.is_cuda and a.flat.numel() == 7_368_769 and float(a.flat.abs().sum()) > 0
        traj[use] = np.array(vals)
    assert np.allclose(traj[False], traj[True], rtol=2e-2), traj          # six optimiser steps: same run-to-run spread


# ---- round 5: which path ran, and the opt-in side stream ------------------------------------------------------------------------------

@pytest.mark.parametrize("cfg_file", ["config_ves_seg-S.yml", "config_ves_seg-S_GAN.yml", "config_gan_ves_seg.yml"])
def test_reference_configs_never_leave_the_hand_written_kernels(cfg_file):
    """The model sections of the three shipped training configs (reference configs/config_ves_seg-S.yml:6-13, -S_GAN.yml, config_gan_ves_seg.yml)
    with `amp: true` as shipped: two training steps and an evaluation pass count ZERO vendor fallbacks and at least one pass on the MFMA /
    exact-fp32 kernels (OCTA_STRICT=1 in tests/conftest.py would also have raised)."""
    import os
    import yaml
    from octa_autosegmentation_amd.models import networks
    from octa_autosegmentation_amd.models.gan_seg_trainer import GanSegTrainer
    from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = yaml.safe_load(open(os.path.join(root, "configs", cfg_file)))
    assert cfg["General"]["amp"] is True
    small = {"General": {"amp": True, "model": cfg["General"]["model"]}, "Train": {k: v for k, v in cfg["Train"].items() if k in ("lr", "loss", "loss_dg", "loss_s", "epochs", "epochs_decay")}}
    before = dict(networks.PATH_COUNTS)
    torch.manual_seed(0)
    if cfg["General"]["model"]["name"] == "GanSegModel":
        small["General"]["model"] = {**cfg["General"]["model"], "upshape": (256, 256)}
        tr = GanSegTrainer(small, "cuda", upshape=(256, 256))
        batch = {"real_A": torch.rand(2, 1, 64, 64, device="cuda"), "real_B": torch.rand(2, 1, 64, 64, device="cuda"),
                 "real_A_seg": (torch.rand(2, 1, 256, 256, device="cuda") > 0.7).float()}
        for _ in range(2):
            tr.perform_training_step(batch)
        with torch.no_grad():
            tr.impl.eval()
            tr.impl(torch.rand(1, 1, 64, 64, device="cuda"))                    # fp32 inference of the segmentor: csrc/conv_f32.hip
    else:
        tr = SegmentationTrainer(small, "cuda")
        x, y = torch.rand(2, 1, 256, 256, device="cuda"), (torch.rand(2, 1, 256, 256, device="cuda") > 0.7).float()
        for _ in range(2):
            tr.perform_training_step({"image": x, "label": y})
        with torch.no_grad():
            tr.impl.eval()
            tr.impl(x[:1])
    torch.cuda.synchronize()
    after = networks.PATH_COUNTS
    assert after["vendor"] == before.get("vendor", 0), "a pass of a shipped config fell back to the torch modules"
    assert after["mfma"] > before.get("mfma", 0) and after["f32_mfma"] > before.get("f32_mfma", 0)


def test_weight_gradients_added_straight_to_the_gradient_arena(monkeypatch):
    """Round 5 (models/mfma_conv.py direct_weight_grads): inside the trainers' backward scope the 3x3 weight-gradient launches add their
    result to `weight.grad` in the parameter layout (octa_conv3x3_nhwc_wgrad_acc) and report None to autograd -- same kernels, same
    operands: every parameter gradient equals the pass that hands autograd one tensor per layer up to the summation order of the
    partial tiles (1e-4 of the tensor's scale), a second backward accumulates in place, and the convolutions really took that route."""
    from octa_autosegmentation_amd.models import mfma_conv
    from octa_autosegmentation_amd.models.segmentation_trainer import IDENTITY_POST, SegmentationTrainer
    from octa_autosegmentation_amd.utils.enums import Phase
    cfg = {"General": {"amp": True, "model": {"name": "DynUNet", "spatial_dims": 2, "in_channels": 1, "out_channels": 1, "kernel_size": [3, 3, 3, 3, 3],
                                               "strides": [1, 2, 2, 2, 1], "upsample_kernel_size": [1, 2, 2, 2, 1]}},
           "Train": {"lr": 1e-3, "loss": "DiceBCELoss", "epochs": 10, "epochs_decay": 0}}
    x, y = torch.rand(2, 1, 128, 160, device="cuda"), (torch.rand(2, 1, 128, 160, device="cuda") > 0.7).float()
    torch.manual_seed(5)
    tr = SegmentationTrainer(cfg, "cuda")
    grads, took = [], []
    for direct in (False, True, True):
        monkeypatch.setattr(mfma_conv, "USE_DIRECT_WGRAD", direct)
        if len(grads) < 2:
            tr.impl.zero_grads("optimizer")
        before = list(mfma_conv.DIRECT_WGRAD_COUNTS)
        with tr.impl.autocast():
            _, losses = tr.impl.inference({"image": x, "label": y}, IDENTITY_POST, torch.device("cuda"), phase=Phase.TRAIN)
            loss = sum(losses.values())
        with tr.impl.backward_scope():
            loss.backward()
        torch.cuda.synchronize()
        took.append(mfma_conv.DIRECT_WGRAD_COUNTS[0] - before[0])
        grads.append({k: p.grad.detach().clone() for k, p in tr.model.named_parameters()})
    assert took[0] == 0 and took[1] == took[2] == 17, took        # the 3x3 layers with 32-multiple channels: 9 encoder + 8 decoder (the one-channel first layer goes through autograd)
    for k, g0 in grads[0].items():
        scale = g0.abs().max().item() + 1e-12
        assert (grads[1][k] - g0).abs().max().item() <= 1e-4 * scale, k                 # in place == one tensor per layer
        assert (grads[2][k] - 2 * g0).abs().max().item() <= 2e-4 * scale, k             # the third backward accumulated onto the second
