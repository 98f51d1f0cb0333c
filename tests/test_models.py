"""CPU: DynUNet restatement (shape, parameter count, MONAI-style state_dict keys), losses, and the
data-parallel gradient exchange over gloo with two processes."""
import os
import socket

import numpy as np
import pytest
import torch

from octa_autosegmentation_amd.models import losses, networks

CFG = {"General": {"amp": True, "model": {"name": "DynUNet", "spatial_dims": 2, "in_channels": 1, "out_channels": 1,
                                           "kernel_size": [3, 3, 3, 3, 3], "strides": [1, 2, 2, 2, 1],
                                           "upsample_kernel_size": [1, 2, 2, 2, 1]}},
       "Train": {"lr": 1e-4, "loss": "DiceBCELoss", "epochs": 30, "epochs_decay": 10}}


def test_dynunet_shape_params_and_keys():
    kw = dict(CFG["General"]["model"]); kw.pop("name")
    net = networks.DynUNet(**kw)
    assert sum(p.numel() for p in net.parameters()) == 7_368_769
    x = torch.randn(2, 1, 64, 64)
    assert net(x).shape == (2, 1, 64, 64)
    sd = net.state_dict()
    for k in ("input_block.conv1.conv.weight", "downsamples.2.norm2.bias", "bottleneck.conv2.conv.weight",
              "upsamples.0.transp_conv.conv.weight", "upsamples.3.conv_block.conv1.conv.weight",
              "output_block.conv.conv.bias", "skip_layers.downsample.conv1.conv.weight",
              "skip_layers.next_layer.next_layer.next_layer.next_layer.conv1.conv.weight",
              "skip_layers.upsample.transp_conv.conv.weight"):
        assert k in sd, k
    assert sd["upsamples.0.transp_conv.conv.weight"].shape == (512, 256, 1, 1)
    assert sd["upsamples.3.transp_conv.conv.weight"].shape == (64, 32, 2, 2)
    net2 = networks.DynUNet(**kw)
    net2.load_state_dict(sd)            # aliases are ignored on load
    assert torch.equal(net2(x), net(x))


def test_dice_bce_formula():
    torch.manual_seed(0)
    logits, y = torch.randn(3, 1, 16, 16), (torch.rand(3, 1, 16, 16) > 0.7).float()
    p = torch.sigmoid(logits)
    dice = torch.mean(torch.stack([1 - (2 * (p[b] * y[b]).sum() + 1e-5) / (p[b].sum() + y[b].sum() + 1e-5) for b in range(3)]))
    want = (dice + torch.nn.functional.binary_cross_entropy_with_logits(logits, y)) / 2
    assert torch.allclose(losses.DiceBCELoss(True)(logits, y), want, atol=1e-7)


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                      # different init per rank: must be overwritten by rank 0's
    tr = SegmentationTrainer(CFG, "cpu", channels_last=False)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(2, 1, 32, 32, generator=g); y = (torch.rand(2, 1, 32, 32, generator=g) > 0.6).float()
    tr.perform_training_step({"image": x[rank:rank + 1], "label": y[rank:rank + 1]})
    if rank == 0:
        torch.save({k: v.clone() for k, v in tr.model.state_dict().items() if not k.startswith("skip_layers")}, out)
    dist.destroy_process_group()


def test_two_rank_data_parallel_step_equals_single_process_batch(tmp_path):
    import torch.multiprocessing as mp
    from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rank0.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    torch.manual_seed(100)
    ref = SegmentationTrainer(CFG, "cpu", channels_last=False)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(2, 1, 32, 32, generator=g); y = (torch.rand(2, 1, 32, 32, generator=g) > 0.6).float()
    # mean over ranks of per-rank (batch 1) gradients == gradient of the mean of the two per-sample losses
    ref.optimizer.zero_grad()
    l = sum(ref.loss_function(ref.model(x[i:i + 1]), y[i:i + 1]) for i in range(2)) / 2
    l.backward(); ref.optimizer.step()
    for k, v in ref.model.state_dict().items():
        if k.startswith("skip_layers"):
            continue
        assert torch.allclose(got[k], v, atol=1e-6), k


def test_gan_networks_shapes_params_and_keys():
    g, d = networks.resnetGenerator9(), networks.patchGAN70x70()
    assert sum(p.numel() for p in g.parameters()) == 11_365_633
    assert sum(p.numel() for p in d.parameters()) == 2_762_689
    x = torch.rand(1, 1, 64, 64)
    y = g(x)
    assert y.shape == x.shape and float(y.min()) >= 0 and float(y.max()) <= 1
    assert networks.patchGAN70x70()(torch.rand(1, 1, 304, 304)).shape == (1, 1, 36, 36)
    sd = g.state_dict()
    for k in ("model.1.weight", "model.1.bias", "model.7.filt", "model.12.conv_block.1.weight", "model.20.conv_block.5.bias",
              "model.21.filt", "model.22.weight", "model.30.weight"):
        assert k in sd, k
    assert "model.2.filt" in d.state_dict() and "model.11.weight" in d.state_dict()
    # blur filters: 3-tap down (sum 1), 4-tap up (sum stride^2)
    assert abs(float(sd["model.7.filt"][0].sum()) - 1) < 1e-6 and abs(float(sd["model.21.filt"][0].sum()) - 4) < 1e-6


def test_gan_seg_training_step_runs_on_cpu():
    from octa_autosegmentation_amd.models.gan_seg_trainer import GanSegTrainer
    cfg = {"General": {"amp": False, "model": {"name": "GanSegModel", "model_g": {"name": "resnetGenerator9"},
                                                 "model_d": {"name": "patchGAN70x70"},
                                                 "model_s": dict(CFG["General"]["model"]), "upshape": (64, 64)}},
           "Train": {"lr": 2e-4, "loss_dg": "LSGANLoss", "loss_s": "DiceBCELoss"}}
    torch.manual_seed(0)
    tr = GanSegTrainer(cfg, "cpu")
    batch = {"real_A": torch.rand(1, 1, 32, 32), "real_B": torch.rand(1, 1, 32, 32), "real_A_seg": (torch.rand(1, 1, 64, 64) > 0.7).float()}
    w0 = tr.segmentor.output_block.conv.conv.weight.clone()
    out, losses = tr.perform_training_step(batch)
    assert set(losses) == {"S", "D_fake", "D_real", "G", "G_idt", "S_idt"} and all(torch.isfinite(v) for v in losses.values())
    assert not torch.equal(w0, tr.segmentor.output_block.conv.conv.weight)
    assert out["prediction"].shape == (1, 1, 64, 64)


def test_lr_schedulers_attach_like_the_reference_by_default():
    """reference models/base_model_abc.py:63-64: every LambdaLR is built on the LAST optimiser (stale loop variable), so with
    Train.epochs_decay > 0 only optimizer_S of the GAN-seg model decays; Train.lr_scheduler_per_optimizer: true is the other behaviour."""
    from octa_autosegmentation_amd.models.gan_seg_trainer import GanSegTrainer
    base = {"General": {"amp": False, "model": {"name": "GanSegModel", "model_g": {"name": "resnetGenerator9"}, "model_d": {"name": "patchGAN70x70"},
                                                  "model_s": dict(CFG["General"]["model"]), "upshape": (64, 64)}},
            "Train": {"lr": 2e-4, "loss_dg": "LSGANLoss", "loss_s": "DiceBCELoss", "epochs": 4, "epochs_decay": 2}}
    lr = lambda o: o.param_groups[0]["lr"]
    tr = GanSegTrainer(base, "cpu")
    assert all(s.optimizer is tr.optimizer_S for s in tr.lr_schedulers) and len(tr.lr_schedulers) == 3
    for _ in range(3):
        for o in (tr.optimizer_G, tr.optimizer_D, tr.optimizer_S):
            o.step()
        for s in tr.lr_schedulers:
            s.step()
    assert lr(tr.optimizer_G) == lr(tr.optimizer_D) == 2e-4 and abs(lr(tr.optimizer_S) - 1e-4) < 1e-12     # epoch 3 of 4, decay over 2
    cfg2 = {**base, "Train": {**base["Train"], "lr_scheduler_per_optimizer": True}}
    tr2 = GanSegTrainer(cfg2, "cpu")
    assert [s.optimizer for s in tr2.lr_schedulers] == [tr2.optimizer_G, tr2.optimizer_D, tr2.optimizer_S]
    for _ in range(3):
        for o in (tr2.optimizer_G, tr2.optimizer_D, tr2.optimizer_S):
            o.step()
        for s in tr2.lr_schedulers:
            s.step()
    assert all(abs(lr(o) - 1e-4) < 1e-12 for o in (tr2.optimizer_G, tr2.optimizer_D, tr2.optimizer_S))
    # a save / resume of the multi-optimiser model under both settings (advisor, round 5): optimiser state comes back per optimiser, the
    # schedulers are NOT part of a checkpoint (as in the reference) and re-attach by the setting of the resuming run
    import tempfile
    from types import SimpleNamespace
    from octa_autosegmentation_amd.utils import checkpoints
    from octa_autosegmentation_amd.utils.enums import Phase
    with tempfile.TemporaryDirectory() as d:
        checkpoints.save_epoch(d, tr, 2, base, save_interval=10, save_best=False)
        for per in (False, True):
            cfg3 = {"General": dict(base["General"]), "Train": {**base["Train"], "lr_scheduler_per_optimizer": per}, "Output": {"save_dir": d}}
            tr3 = GanSegTrainer(cfg3, "cpu", args=SimpleNamespace(start_epoch=3, epoch="latest"))
            want = [tr3.optimizer_G, tr3.optimizer_D, tr3.optimizer_S] if per else [tr3.optimizer_S] * 3
            assert [s.optimizer for s in tr3.lr_schedulers] == want
            for k, v in tr.generator.state_dict().items():
                assert torch.equal(v, tr3.generator.state_dict()[k]), k


def test_t1x1_packs_follow_the_invalidation_epoch():
    """A write through `.data` moves neither the version counter nor the storage: the cached packs of the 1x1 transposed convolution
    must be rebuilt after invalidate_all_pack_plans() (round-4 advisor item)."""
    from octa_autosegmentation_amd.models import mfma_conv
    w = torch.nn.Parameter(torch.randn(64, 32, 1, 1))
    f0, d0 = mfma_conv._t1x1_packs(w)
    assert mfma_conv._t1x1_packs(w)[0] is f0                         # cached
    w.data.mul_(2.0)
    assert mfma_conv._t1x1_packs(w)[0] is f0                         # nothing moved: stale, as documented
    holder = torch.nn.Module()
    holder.w = w
    mfma_conv.invalidate_all_pack_plans(holder)
    f1, d1 = mfma_conv._t1x1_packs(w)
    assert f1 is not f0 and torch.equal(mfma_conv.tap_major(f1)[4].float(), w.detach().reshape(64, 32).t().to(torch.bfloat16).float())
    assert torch.equal(mfma_conv.tap_major(d1)[4].float(), w.detach().reshape(64, 32).to(torch.bfloat16).float())


def test_noise_transforms_match_the_reference_fixture():
    """a17: SpeckleBrightnesd and AddRandomBackgroundNoised against tests/golden/noise_golden.npz -- outputs of the reference's own
    classes (tools/make_golden_noise.py) -- with the same generator seeds: CPU tensors, bit for bit (same torch / numpy draws, same
    arithmetic, the float64 promotion of the background product included)."""
    from octa_autosegmentation_amd.data import data_transforms as T
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "noise_golden.npz"))
    for k in range(2):
        img = torch.from_numpy(g[f"speckle_{k}_in"])
        torch.manual_seed(100 + k)
        out = T.SpeckleBrightnesd(["image"])({"image": img.clone()})["image"]
        assert out.dtype == torch.float32 and torch.equal(out, torch.from_numpy(g[f"speckle_{k}_out"]))
        np.random.seed(200 + k)
        d = T.AddRandomBackgroundNoised(["image"])({"image": img.clone(), "background": torch.from_numpy(g[f"bg_{k}_noise"])})
        assert "background" not in d and d["image"].dtype == torch.float64 and torch.equal(d["image"], torch.from_numpy(g[f"bg_{k}_out"]))
        np.random.seed(300 + k); torch.manual_seed(300 + k)
        d = T.AddRandomBackgroundNoised(["image"])({"image": img.clone()})
        assert torch.equal(d["image"], torch.from_numpy(g[f"bg_{k}_out_nobg"]))
    keep = T.AddRandomBackgroundNoised(["image"], delete_background=False)({"image": img.clone(), "background": img.clone()})
    assert "background" in keep
    gnet = networks.resnetGenerator9()
    o3 = T.ImageToImageTranslationd(keys=["image"], model=gnet, device="cpu")({"image": img})["image"]
    with torch.no_grad():
        assert torch.equal(o3, gnet.eval()(img.unsqueeze(0)).squeeze(0))


def test_checkpoint_files_round_trip_and_metrics_csv(tmp_path):
    """File names, dict keys and resume order of the reference's trainer."""
    from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
    from octa_autosegmentation_amd.utils import checkpoints
    torch.manual_seed(0)
    a = SegmentationTrainer(CFG, "cpu", channels_last=False)
    x = torch.rand(1, 1, 32, 32); y = (torch.rand(1, 1, 32, 32) > 0.5).float()
    a.perform_training_step({"image": x, "label": y})
    d = str(tmp_path)
    files = checkpoints.save_epoch(d, a, 9, CFG, save_interval=10, save_best=True)
    names = sorted(os.listdir(os.path.join(d, "checkpoints")))
    assert names == sorted(f"{p}_{n}_model.pth" for p in ("latest", "10", "best") for n in ("optimizer", "model"))
    ck = torch.load(files[-1], weights_only=False)
    assert set(ck) == {"epoch", "model", "optimizer", "config"} and ck["epoch"] == 10 and ck["optimizer"] is None
    assert any(k.startswith("skip_layers.") for k in ck["model"])            # MONAI's aliased keys are present
    torch.manual_seed(1)
    b = SegmentationTrainer(CFG, "cpu", channels_last=False)
    assert checkpoints.load_checkpoint(b, os.path.join(d, "checkpoints", "latest_model.pth")) == 10
    for (k, v), (_, w) in zip(a.model.state_dict().items(), b.model.state_dict().items()):
        assert torch.equal(v, w), k
    assert b.optimizer.state_dict()["state"][0]["step"] == a.optimizer.state_dict()["state"][0]["step"]
    log = checkpoints.MetricsLog(d, CFG)
    log.append(0, {"loss": {"train_loss": 0.5, "val_loss": 0.6}, "metric": {"val_DSC": 0.7}})
    log.append(1, {"loss": {"train_loss": 0.4, "val_loss": 0.5}, "metric": {"val_DSC": 0.8}})
    rows = open(os.path.join(d, "metrics.csv")).read().splitlines()
    assert rows[0] == "epoch,train_loss,val_loss,val_DSC" and rows[2] == "1,0.4,0.5,0.8"
    assert os.path.exists(os.path.join(d, "config.yml"))


def _gan_worker(rank, world, port, outdir):
    import torch.distributed as dist
    from octa_autosegmentation_amd.models.gan_seg_trainer import GanSegTrainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = {"General": {"amp": False, "model": {"name": "GanSegModel", "model_g": {"name": "resnetGenerator9"},
                                                 "model_d": {"name": "patchGAN70x70"},
                                                 "model_s": dict(CFG["General"]["model"]), "upshape": (64, 64)}},
           "Train": {"lr": 2e-4, "loss_dg": "LSGANLoss", "loss_s": "DiceBCELoss"}}
    torch.manual_seed(200 + rank)                      # different init and different data per rank
    tr = GanSegTrainer(cfg, "cpu")
    batch = {"real_A": torch.rand(1, 1, 32, 32), "real_B": torch.rand(1, 1, 32, 32), "real_A_seg": (torch.rand(1, 1, 64, 64) > 0.7).float()}
    tr.perform_training_step(batch)
    state = {f"{n}.{k}": v.clone() for n, m in (("g", tr.generator), ("d", tr.discriminator), ("s", tr.segmentor))
             for k, v in m.state_dict().items() if "skip_layers" not in k}
    torch.save(state, os.path.join(outdir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_gan_seg_step_keeps_the_replicas_identical(tmp_path):
    """GAN-seg trainer under torch.distributed (gloo, 2 ranks, different data and different seeds per rank): rank 0's parameters are
    broadcast at construction and every optimiser sees the rank-averaged gradients, so G, D and S are identical on both ranks after
    the step (the path bench.py --gpus N and train_synthetic.py --gan run over RCCL)."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_gan_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(str(tmp_path / "rank0.pt")), torch.load(str(tmp_path / "rank1.pt"))
    assert a.keys() == b.keys() and len(a) > 100
    for k in a:
        assert torch.equal(a[k], b[k]), k


def run_gan_seg_fixture(tag, idt, device="cpu", amp=False, arena=False):
    """Two consecutive joint G / D / S updates of this repository's GanSegModel on the closed-form weights and inputs of
    tools/make_golden_ganseg.py; returns (losses [2, 6], gradient norms [3], parameter checksums [3, 2], the fixture)."""
    import sys
    from argparse import Namespace
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from tools.make_golden_ganseg import S_CFG, TRAIN, WEIGHTS, batch, checksums, grad_norms
    from octa_autosegmentation_amd.models.model import define_model
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ganseg_golden.npz"))
    config = {"General": {"device": device, "amp": amp, "model": {"name": "GanSegModel", "model_g": {"name": "resnetGenerator9"},
                                                                   "model_d": {"name": "patchGAN70x70"}, "model_s": dict(S_CFG),
                                                                   "compute_identity": idt, "compute_identity_seg": True, "upshape": (64, 64)}},
              "Train": dict(TRAIN), "Output": {"save_dir": "/tmp"}}
    from copy import deepcopy
    from octa_autosegmentation_amd.utils.enums import Phase
    torch.manual_seed(0)
    old = os.environ.get("OCTA_GRAD_ARENA")
    if arena:
        os.environ["OCTA_GRAD_ARENA"] = "1"
    try:
        model = define_model(deepcopy(config), Phase.TRAIN)
        model.initialize_model_and_optimizer(None, networks.init_weights, config, Namespace(start_epoch=0, epoch="latest"), None, Phase.TRAIN)
    finally:
        if arena:
            os.environ.pop("OCTA_GRAD_ARENA") if old is None else os.environ.__setitem__("OCTA_GRAD_ARENA", old)
    assert bool(model._arenas) == arena
    for salt, name in ((0, "generator"), (100, "discriminator"), (200, "segmentor")):
        WEIGHTS["he_" if tag.startswith("he_") else ""](getattr(model, name), salt)
    model._after_weight_surgery()
    model.train()
    ident = {"prediction": lambda t: t, "label": lambda t: t}
    keys = ("S", "D_fake", "D_real", "G", "G_idt", "S_idt")
    losses, norms = [], []
    for step in range(2):
        _, l = model.perform_training_step(batch(), None, ident, device)
        losses.append([float(l[k]) for k in keys])
        norms.append(grad_norms(model))
    return np.array(losses), np.array(norms), checksums(model), g


@pytest.mark.parametrize("tag,idt", [("idt0", False), ("idt1", True), ("he_idt0", False), ("he_idt1", True)])
def test_gan_seg_update_matches_the_reference_fixture(tag, idt):
    """a21: two consecutive joint G / D / S updates against tests/golden/ganseg_golden.npz, recorded from the reference's own
    GanSegModel.perform_training_step (tools/make_golden_ganseg.py). Losses of step 1 depend only on the forward composition,
    those of step 2 on all three optimiser updates (Adam, betas (0.5, 0.999) for G and D, (0.9, 0.999) for S); the gradient norms and
    parameter checksums pin the backward paths (D frozen in the G+S pass, detached fake_B in the D pass, detached pseudo-labels).
    `idt*`: parameters from the `fill` ramp; `he_idt*` (round 3): well-conditioned He-scaled parameters from numpy's legacy generator
    (tools/make_golden_ganseg.he_weights) -- the variant that also runs on the GPU in fp32 and through the bf16 / MFMA path
    (tests/test_models_gpu.py); with the ramp some InstanceNorm inputs are nearly constant, the generator's gradient norm is 1e6 and
    a different summation order alone moves step-2 losses by tens of percent."""
    losses, gnorm, sums, g = run_gan_seg_fixture(tag, idt)
    # step 0: the forward composition alone (fp32 summation order); step 1: after one Adam step of all three optimisers, where
    # first-step updates are lr * sign(g) and parameters with rounding-noise gradients move either way
    for step in range(2):
        assert np.allclose(losses[step], g[f"{tag}_losses"][step], rtol=2e-5 if step == 0 else 1e-3, atol=1e-6), (step, losses[step], g[f"{tag}_losses"][step])
    assert np.allclose(gnorm, g[f"{tag}_grad_norms_steps"], rtol=2e-3), (gnorm, g[f"{tag}_grad_norms_steps"])
    # parameter checksums after two Adam steps: parameters whose gradient is rounding noise take +-lr steps of either sign, so the
    # sums agree to a small absolute slack only (the step-2 losses above are the sharp pin of the three updates)
    assert np.allclose(sums, g[f"{tag}_param_sums"], rtol=1e-6, atol=0.5), (sums, g[f"{tag}_param_sums"])
    # and a sign error would not pass: the generator's adversarial term enters loss_GS with a plus sign
    assert g[f"{tag}_losses"][0][3] > 1.0


def test_batched_translation_chain_takes_the_per_sample_decisions():
    """configs/config_ves_seg-S_GAN.yml's loader chain cut at the frozen generator (Compose.call_batch): prefixes, ONE generator pass
    over the mini-batch, suffixes -- same tensors and same stream positions as sample by sample (reference data_transforms.py:327-356
    runs the generator once per sample inside the loader workers)."""
    import random
    from octa_autosegmentation_amd.data import data_transforms as T
    from octa_autosegmentation_amd.data.image_dataset import ListDataset
    torch.manual_seed(3)
    gnet = networks.resnetGenerator9()
    networks.init_weights(gnet, "kaiming", nonlinearity="relu")
    aug = [{"name": "AddRandomBackgroundNoised", "keys": ["image"], "delete_background": False},
           {"name": "ImageToImageTranslationd", "keys": ["image"], "model": gnet, "device": "cpu"},
           {"name": "SpeckleBrightnesd", "keys": ["image"]},
           {"name": "RandFlipd", "keys": ["image", "label"], "prob": 0.5, "spatial_axis": [0, 1]},
           {"name": "RandRotate90d", "keys": ["image", "label"], "prob": 0.75}]
    items = [{"image": torch.rand(1, 32, 32), "label": torch.rand(1, 32, 32), "background": torch.rand(1, 32, 32)} for _ in range(3)]
    got = {}
    for batched in (True, False):
        ds = ListDataset(items, T.Compose(T.get_data_augmentations(aug, seed=5)))
        assert ds.transform.batchable()
        random.seed(1); np.random.seed(1); torch.manual_seed(1)
        out = ds.get_batch([0, 1, 2]) if batched else [ds[i] for i in range(3)]
        got[batched] = (out, random.random(), np.random.random_sample(), float(torch.rand(())))
    assert got[True][1:] == got[False][1:]
    for a, b in zip(got[True][0], got[False][0]):
        assert torch.allclose(a["image"], b["image"], atol=1e-6) and torch.equal(a["label"], b["label"])
    # without a background tile the noise stand-in draws from torch's stream, which the speckle transform behind the cut shares:
    # such mini-batches stay sample by sample
    bare = [{k: v for k, v in it.items() if k != "background"} for it in items]
    nob = ListDataset(bare, T.Compose(T.get_data_augmentations(aug, seed=5)))
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    out = nob.get_batch([0, 1, 2])
    nob = ListDataset(bare, T.Compose(T.get_data_augmentations(aug, seed=5)))
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    ref = [nob[i] for i in range(3)]
    assert all(torch.equal(a["image"], b["image"]) for a, b in zip(out, ref))
    assert not nob.transform.batchable(has_background=False) and nob.transform.batchable(has_background=True)
    # a chain with the speckle transform (torch's stream) IN FRONT of the cut and another torch user behind it must not be reordered:
    # the guard reads the transforms' declared streams (round 3 asked for an attribute no transform defined)
    swapped = [aug[0], aug[2], aug[1], {"name": "SpeckleBrightnesd", "keys": ["image"]}] + aug[3:]
    sw = ListDataset(items, T.Compose(T.get_data_augmentations(swapped, seed=5)))
    assert not sw.transform.batchable()
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    out = sw.get_batch([0, 1, 2])
    sw = ListDataset(items, T.Compose(T.get_data_augmentations(swapped, seed=5)))
    random.seed(1); np.random.seed(1); torch.manual_seed(1)
    ref = [sw[i] for i in range(3)]
    assert all(torch.equal(a["image"], b["image"]) for a, b in zip(out, ref))
    # one torch user in front of the cut and none behind it is fine
    front = [aug[0], aug[2], aug[1]] + aug[3:]
    assert ListDataset(items, T.Compose(T.get_data_augmentations(front, seed=5))).transform.batchable()


def test_f32_weight_pack_cache_cannot_serve_a_collected_parameters_copy():
    """models/conv_f32.py caches the kernel's weight layout. Keyed by id(weight) + version + storage address it served a STALE copy when
    a new parameter was given the id and the address of a collected one (two networks built one after the other; found as a flaky
    full-size logits test). The cache now lives on the parameter object: build, pack and drop parameters in a loop -- CPython and the
    allocator reuse ids and addresses readily -- and every pack must be the current parameter's."""
    import gc
    import torch
    from octa_autosegmentation_amd.models import conv_f32
    reused = 0
    seen = set()
    for k in range(40):
        w = torch.nn.Parameter(torch.full((4, 3, 3, 3), float(k)))
        key = (id(w), w.data_ptr())
        reused += key in seen
        seen.add(key)
        p = conv_f32._packed(w, False)
        assert p.shape == (3, 9, 4) and float(p.min()) == float(p.max()) == float(k), k
        pt = conv_f32._packed(w, True)
        assert pt.shape == (4, 9, 3) and float(pt.min()) == float(k)
        assert conv_f32._packed(w, False) is not pt            # the orientation is part of the key
        with torch.no_grad():
            w.add_(1.0)                                        # version bump: repacked
        assert float(conv_f32._packed(w, False).max()) == float(k + 1)
        w.data.fill_(-1.0)                                     # no version bump: callers invalidate
        conv_f32.invalidate_packs()
        assert float(conv_f32._packed(w, False).max()) == -1.0
        del w, p, pt
        gc.collect()
    # (`reused` usually is > 0 here -- the id-keyed cache failed this loop whenever it was; whether the interpreter and the allocator
    # reuse an id together with an address depends on the process's history, so it is not asserted)
