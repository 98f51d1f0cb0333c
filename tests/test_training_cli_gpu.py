"""GPU: the reference's entry points end to end on MI355X with the reference's own config files -- generate_vessel_graph.py ->
train.py (configs/config_ves_seg-S.yml, then configs/config_gan_ves_seg.yml) -> test.py -> validate.py -- plus the graph loader
transform (a16) against the general rasterize_forest path."""
import glob
import os
import random
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def graphs(tmp_path_factory, hip_lib_built):
    """Eight short seeded samples written by the generator CLI: graph CSV + 304x304 image + 1216x1216 label each."""
    import generate_vessel_graph
    from octa_autosegmentation_amd.utils import configs
    cfg = configs.load_generator_config()
    out = tmp_path_factory.mktemp("graphs")
    modes = [dict(cfg["Greenhouse"]["modes"][0], I=30), dict(cfg["Greenhouse"]["modes"][1], I=20)]
    generate_vessel_graph.main(["--config_file", configs.GENERATOR_CONFIG, "--num_samples", "8", "--seed", "100", "--labels",
                                "--output.directory", str(out), "--Greenhouse.modes", yaml.safe_dump(modes, default_flow_style=True).strip()])
    dirs = sorted(glob.glob(str(out / "*")))
    assert len(dirs) == 8
    return str(out), dirs


def test_generator_cli_writes_complete_triples(graphs):
    from PIL import Image
    from octa_autosegmentation_amd import graph_io
    from oracle import octa_oracle
    out, dirs = graphs
    for d in dirs[:2]:
        name = os.path.basename(d)
        e = graph_io.read_csv(os.path.join(d, name + ".csv"))
        assert (graph_io.read_csv_native(os.path.join(d, name + ".csv")) == e).all()
        label = Image.open(os.path.join(d, name + "_label.png"))
        assert label.mode == "1" and label.size == (1216, 1216)
        want = octa_oracle.fs_dither(octa_oracle.rasterize(e, [1216, 1216]))          # the label IS the raster of the file's text
        assert (np.array(label.convert("L")) == want).all()
        img = np.array(Image.open(os.path.join(d, "art_ven_img_gray.png")))
        assert img.shape == (304, 304) and img.dtype == np.uint8 and img.max() > 0


def test_graph_loader_transform_fast_path_equals_reference_path(graphs):
    """a16: LoadGraphAndFilterByRandomRadiusd on a CSV file -- the native-reader / device-window fast path gives the tensors and
    consumes the `random` stream exactly like the general path (csv.DictReader + rasterize_forest with string positions, the
    reference's data_transforms.py:369-387), incl. the shared blackdict of the second key; and both equal the oracle."""
    import torch
    from octa_autosegmentation_amd import graph_io
    from octa_autosegmentation_amd.data import data_transforms as T
    from octa_autosegmentation_amd.vessel_graph_generation.tree2img import rasterize_forest
    from oracle import octa_oracle
    import csv
    out, dirs = graphs
    path = glob.glob(os.path.join(dirs[0], "*.csv"))[0]
    t = T.LoadGraphAndFilterByRandomRadiusd(["image", "label"], image_resolutions=[[304, 304], [1216, 1216]], min_radius=[0, 0.0033], max_dropout_prob=0)
    random.seed(7)
    got = t({"image": path, "label": path})
    after_fast = random.random()
    random.seed(7)
    with open(path, newline="") as f:
        forest = list(csv.DictReader(f))
    img, bd = rasterize_forest(forest, [304, 304], 2, min_radius=0, max_dropout_prob=0, blackdict=None)
    lab, bd = rasterize_forest(forest, [1216, 1216], 2, min_radius=0.0033, max_dropout_prob=0, blackdict=bd)
    after_ref = random.random()
    assert after_fast == after_ref
    assert got["image"].is_cuda and got["image"].dtype == torch.float32 and tuple(got["label"].shape) == (1216, 1216)
    assert (got["image"].cpu().numpy() == img).all() and (got["label"].cpu().numpy() == lab).all()
    e = graph_io.read_csv(path)
    assert (lab == octa_oracle.rasterize(e[e[:, 6] >= 0.0033], [1216, 1216])).all()
    # with dropout the transform goes through the general path: same blackdict for both keys
    t2 = T.LoadGraphAndFilterByRandomRadiusd(["image", "label"], image_resolutions=[[304, 304], [1216, 1216]], min_radius=[0, 0], max_dropout_prob=0.9)
    random.seed(11)
    d2 = t2({"image": path, "label": path})
    random.seed(11)
    img2, bd2 = rasterize_forest(forest, [304, 304], 2, min_radius=0, max_dropout_prob=0.9, blackdict=None)
    lab2, _ = rasterize_forest(forest, [1216, 1216], 2, min_radius=0, max_dropout_prob=0.9, blackdict=bd2)
    assert (d2["image"].cpu().numpy() == img2).all() and (d2["label"].cpu().numpy() == lab2).all()


def test_fused_batch_loader_equals_the_per_sample_transform_chain(graphs):
    """configs/config_ves_seg-S.yml's training chain: the batch-level loader (two rasteriser launch sequences + two augmentation
    kernels per tensor and mini-batch) against the generic per-sample transforms of the registry, same seed, same shuffled order:
    same random decisions, images within 1e-3 of [0, 1] values (fp32 rounding of the sampling positions), labels equal except pixels within rounding of the 0.1 threshold; the
    global `random` stream ends in the same state."""
    import torch
    from octa_autosegmentation_amd.data.image_dataset import get_dataset
    out, dirs = graphs
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "config_ves_seg-S.yml")))
    csvs = os.path.join(out, "**", "*.csv")
    cfg["Train"]["data"] = {"image": {"files": csvs}, "label": {"files": csvs}}
    cfg["General"].update(amp=False, seed=11)
    got = {}
    for generic in (False, True):
        cfg["General"]["generic_loader"] = generic
        torch.manual_seed(5); random.seed(5)
        loader = get_dataset(cfg, "Train", num_workers=0)
        assert (loader.fused is None) == generic and len(loader) == 2
        got[generic] = (list(loader), random.random())
    (fused, r_f), (plain, r_p) = got[False], got[True]
    assert r_f == r_p
    for a, b in zip(fused, plain):
        assert a["image_path"] == b["image_path"] and a["label_path"] == b["label_path"]
        assert a["image"].shape == b["image"].shape == (4, 1, 1216, 1216) and a["image"].dtype == b["image"].dtype == torch.float32
        assert (a["image"] - b["image"]).abs().max().item() <= 1e-3          # float32 angle and folded intensity map: sub-pixel-shift rounding on steep vessel edges
        assert (a["label"] != b["label"]).float().mean().item() < 1e-4 and set(a["label"].unique().tolist()) <= {0.0, 1.0}
        assert 0.01 < a["label"].mean().item() < 0.5


def _pngs(graph_dirs, root):
    """Validation / test / real_B / background stand-ins: the generated 304x304 images and 1216x1216 labels as flat PNG folders."""
    from PIL import Image
    for sub in ("images", "labels", "background"):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    rng = np.random.default_rng(0)
    for i, d in enumerate(graph_dirs):
        name = os.path.basename(d)
        Image.open(os.path.join(d, "art_ven_img_gray.png")).save(os.path.join(root, "images", f"{i}.png"))
        Image.open(os.path.join(d, name + "_label.png")).convert("L").save(os.path.join(root, "labels", f"{i}.png"))
        Image.fromarray(rng.integers(0, 80, (304, 304)).astype(np.uint8)).save(os.path.join(root, "background", f"{i}.png"))


def test_train_test_validate_with_the_reference_segmentation_config(graphs, tmp_path):
    """BASELINE configs[2] plumbing: train.py --config_file configs/config_ves_seg-S.yml (one epoch on the eight graphs), then
    test.py and validate.py on its checkpoints. Only paths and the epoch count are overridden."""
    import torch
    from PIL import Image
    import test as test_cli
    import train as train_cli
    import validate as validate_cli
    out, dirs = graphs
    data = str(tmp_path / "png")
    _pngs(dirs, data)
    res = str(tmp_path / "results")
    csvs = os.path.join(out, "**", "*.csv")
    ov = ["--Train.data.image.files", csvs, "--Train.data.label.files", csvs, "--Train.epochs", "1", "--Train.epochs_decay", "0",
          "--Validation.data.image", yaml.safe_dump({"files": os.path.join(data, "images", "*.png")}, default_flow_style=True).strip(),
          "--Validation.data.label", yaml.safe_dump({"files": os.path.join(data, "labels", "*.png")}, default_flow_style=True).strip(),
          "--Test.data.image", yaml.safe_dump({"files": os.path.join(data, "images", "*.png")}, default_flow_style=True).strip(),
          "--Output.save_dir", res, "--General.seed", "3"]
    run = train_cli.main(["--config_file", os.path.join(ROOT, "configs", "config_ves_seg-S.yml")] + ov)
    rows = open(os.path.join(run, "metrics.csv")).read().splitlines()
    assert rows[0].startswith("epoch,train_DiceBCELoss,val_DiceBCELoss,Train_DSC,Train_IoU,Validation_DSC,Validation_IoU") and len(rows) == 2
    vals = dict(zip(rows[0].split(","), (float(v) for v in rows[1].split(","))))
    assert np.isfinite(list(vals.values())).all() and 0 < vals["train_DiceBCELoss"] < 2
    names = set(os.listdir(os.path.join(run, "checkpoints")))
    assert {"latest_model_model.pth", "latest_optimizer_model.pth", "best_model_model.pth", "best_optimizer_model.pth"} <= names
    ck = torch.load(os.path.join(run, "checkpoints", "best_model_model.pth"), weights_only=False)
    assert ck["epoch"] == 1 and "input_block.conv1.conv.weight" in ck["model"] and any(k.startswith("skip_layers.") for k in ck["model"])
    run_cfg = os.path.join(run, "config.yml")
    written = test_cli.main(["--config_file", run_cfg, "--epoch", "best", "--num_samples", "3"])
    assert len(written) == 3
    pred = np.array(Image.open(written[0]))
    assert pred.shape == (1216, 1216) and set(np.unique(pred)) <= {0, 255}
    # RemoveSmallObjects(160) ran: no 4-connected component below 160 pixels survives in the written file
    from scipy import ndimage
    lab, n = ndimage.label(pred > 0)
    assert n == 0 or np.bincount(lab.ravel())[1:].min() >= 160
    metrics = validate_cli.main(["--config_file", run_cfg, "--epoch", "best"])
    assert {"Validation_DSC", "Validation_IoU", "Validation_AUC", "Validation_ACC", "Validation_Recall", "Validation_Precision"} <= set(metrics)
    assert all(np.isfinite(v) for v in metrics.values())


@pytest.fixture(scope="module")
def gan_run(graphs, tmp_path_factory):
    """One epoch of train.py --config_file configs/config_gan_ves_seg.yml on the eight graphs (shared by the tests below)."""
    import train as train_cli
    out, dirs = graphs
    tmp = tmp_path_factory.mktemp("ganrun")
    data = str(tmp / "png")
    _pngs(dirs, data)
    res = str(tmp / "results")
    csvs = os.path.join(out, "**", "*.csv")
    f = lambda p: yaml.safe_dump({"files": p}, default_flow_style=True).strip()
    ov = ["--Train.data.real_A", f(csvs), "--Train.data.real_A_seg", f(csvs), "--Train.data.real_B", f(os.path.join(data, "images", "*.png")),
          "--Train.data.background", f(os.path.join(data, "background", "*.png")), "--Train.epochs", "1", "--Train.batch_size", "2",
          "--Output.save_dir", res, "--General.seed", "4"]
    run = train_cli.main(["--config_file", os.path.join(ROOT, "configs", "config_gan_ves_seg.yml")] + ov)
    return run, data, csvs


def test_train_with_the_reference_gan_seg_config(gan_run):
    """BASELINE configs[3] plumbing: train.py --config_file configs/config_gan_ves_seg.yml for one epoch (GanSegModel through
    define_model, UnalignedZipDataset pairing, dropout 0.02 in the graph loader, three optimisers' checkpoints)."""
    import torch
    run, data, csvs = gan_run
    rows = open(os.path.join(run, "metrics.csv")).read().splitlines()
    assert rows[0] == "epoch,train_S,train_D_fake,train_D_real,train_G,train_G_idt,train_S_idt,Train_DSC,Train_IoU" and len(rows) == 2
    assert np.isfinite([float(v) for v in rows[1].split(",")]).all()
    names = set(os.listdir(os.path.join(run, "checkpoints")))
    assert {f"latest_{n}_model.pth" for n in ("generator", "discriminator", "segmentor", "optimizer_G", "optimizer_D", "optimizer_S")} <= names
    ck = torch.load(os.path.join(run, "checkpoints", "latest_optimizer_S_model.pth"), weights_only=False)
    assert ck["optimizer"]["param_groups"][0]["betas"] == (0.9, 0.999) and ck["epoch"] == 1


def test_gan_checkpoints_load_in_test_cli_as_G_and_S(gan_run, tmp_path):
    """train.py writes `<tag>_generator_model.pth` / `<tag>_segmentor_model.pth`; test.py with `General.inference: G` (what
    configs/config_gan_ves_seg.yml ships) and `S` must find them (the reference builds `<tag>_G_model.pth`, which its own trainer
    never writes). G: translated 304x304 images; S: 1216x1216 segmentations of real scans, RemoveSmallObjects applied."""
    from PIL import Image
    import test as test_cli
    run, data, csvs = gan_run
    f = lambda p: yaml.safe_dump({"files": p}, default_flow_style=True).strip()
    cfg = os.path.join(run, "config.yml")
    w = test_cli.main(["--config_file", cfg, "--epoch", "latest", "--num_samples", "2", "--General.inference", "G",
                       "--Test.data", yaml.safe_dump({"real_A": {"files": csvs}, "background": {"files": os.path.join(data, "background", "*.png")}},
                                                     default_flow_style=True).strip(),
                       "--Test.save_dir", str(tmp_path / "G")])
    assert len(w) == 2 and all(os.path.basename(x).startswith("generator_") for x in w)
    assert np.array(Image.open(w[0])).shape == (304, 304)
    w = test_cli.main(["--config_file", cfg, "--epoch", "latest", "--num_samples", "2", "--General.inference", "S",
                       "--Test.data", yaml.safe_dump({"real_B": {"files": os.path.join(data, "images", "*.png")}}, default_flow_style=True).strip(),
                       "--Test.save_dir", str(tmp_path / "S")])
    assert len(w) == 2 and all(os.path.basename(x).startswith("segmentor_") for x in w)
    seg = np.array(Image.open(w[0]))
    assert seg.shape == (1216, 1216) and set(np.unique(seg)) <= {0, 255}


def test_translation_transform_from_a_checkpoint_file_runs_the_mfma_generator(gan_run):
    """a17 ImageToImageTranslationd (reference data/data_transforms.py:327-356) from a checkpoint FILE on the GPU: the frozen
    resnetGenerator9 under bf16 autocast -- MFMA 3x3 stages, thin-conv stems -- one pass per mini-batch, against the CPU fp32 module
    with the same weights. Budget: sigmoid outputs in [0, 1] through 9 residual blocks in bf16: max 0.06, mean 0.01; the batched pass
    equals the per-sample passes to bf16 rounding of identical arithmetic (InstanceNorm is per sample)."""
    import torch
    from octa_autosegmentation_amd.data import data_transforms as T
    from octa_autosegmentation_amd.models import networks
    from octa_autosegmentation_amd.models.base_model_abc import load_checkpoint_file
    run, data, csvs = gan_run
    path = os.path.join(run, "checkpoints", "latest_generator_model.pth")
    t = T.ImageToImageTranslationd(model_path=path, keys=["image"])
    assert t.amp and next(t.model.parameters()).is_cuda and not any(p.requires_grad for p in t.model.parameters())
    g = torch.Generator().manual_seed(0)
    imgs = [torch.rand(1, 304, 304, generator=g).cuda() for _ in range(4)]
    from octa_autosegmentation_amd.models import mfma_conv
    calls, orig, orig_r = [], mfma_conv.conv3x3, mfma_conv.conv3x3_reflect
    mfma_conv.conv3x3 = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    mfma_conv.conv3x3_reflect = lambda *a, **k: (calls.append(2), orig_r(*a, **k))[1]
    try:
        batched = t.batch_apply([{"image": im} for im in imgs])
    finally:
        mfma_conv.conv3x3, mfma_conv.conv3x3_reflect = orig, orig_r
    # 2 down + 2 up 3x3 stages and the 18 convolutions of the residual blocks (reflection fused into their halo fetch) went through the
    # MFMA convolution, once for the mini-batch
    assert calls.count(2) == 18 and len(calls) >= 22
    single = [t({"image": im})["image"] for im in imgs]
    cpu = networks.resnetGenerator9()
    cpu.load_state_dict(load_checkpoint_file(path, "cpu")["model"])
    cpu.eval()
    with torch.no_grad():
        ref = cpu(torch.stack([im.cpu() for im in imgs]))
    for i in range(4):
        b, s1 = batched[i]["image"], single[i]
        assert b.shape == (1, 304, 304) and b.dtype == torch.float32
        assert (b - s1).abs().max().item() <= 1e-2, (b - s1).abs().max().item()
        d = (b.cpu() - ref[i]).abs()
        print(f"[i2i] sample {i}: max {d.max().item():.4f} mean {d.mean().item():.5f}", flush=True)
        assert d.max().item() <= 0.06 and d.mean().item() <= 0.01
    # amp=False: the reference's own arithmetic for the frozen generator (fp32, data_transforms.py:350-356). Round 6: on the exact-fp32 MFMA
    # kernels (csrc/conv_f32.hip: 7x7, 3x3), no vendor kernel (OCTA_STRICT=1 here: a fallback would raise), within 1e-4 of the CPU modules
    t32 = T.ImageToImageTranslationd(model_path=path, keys=["image"], amp=False)
    before = dict(networks.PATH_COUNTS)
    d32 = (t32.batch_apply([{"image": im} for im in imgs])[0]["image"].cpu() - ref[0]).abs().max().item()
    assert networks.PATH_COUNTS["vendor"] == before.get("vendor", 0) and networks.PATH_COUNTS["f32"] > before.get("f32", 0)
    assert d32 <= 1e-4, d32


def test_train_with_the_reference_s_gan_config(graphs, gan_run, tmp_path):
    """BASELINE configs[3] AS NAMED: train.py --config_file configs/config_ves_seg-S_GAN.yml for one epoch -- background blend ->
    frozen generator from the checkpoint the GAN run above wrote -> resize -> speckle -> flips / rotations -> DynUNet-S step. Only
    paths, the epoch count and the generator checkpoint path are overridden. The loader's batched generator pass is checked against
    the per-sample chain on the same seed."""
    import torch
    import train as train_cli
    from octa_autosegmentation_amd.data.image_dataset import get_dataset
    out, dirs = graphs
    run, data, csvs = gan_run
    cfg = yaml.safe_load(open(os.path.join(ROOT, "configs", "config_ves_seg-S_GAN.yml")))
    aug = cfg["Train"]["data_augmentation"]
    i2i = [a for a in aug if a["name"] == "ImageToImageTranslationd"]
    assert len(i2i) == 1
    i2i[0]["model_path"] = os.path.join(run, "checkpoints", "latest_generator_model.pth")
    f = lambda p: yaml.safe_dump({"files": p}, default_flow_style=True).strip()
    res = str(tmp_path / "results")
    ov = ["--Train.data.image", f(csvs), "--Train.data.label", f(csvs), "--Train.data.background", f(os.path.join(data, "background", "*.png")),
          "--Train.data_augmentation", yaml.safe_dump(aug, default_flow_style=True, width=100000).strip(),
          "--Train.epochs", "1", "--Train.epochs_decay", "0",
          "--Validation.data.image", f(os.path.join(data, "images", "*.png")), "--Validation.data.label", f(os.path.join(data, "labels", "*.png")),
          "--Output.save_dir", res, "--General.seed", "5"]
    run2 = train_cli.main(["--config_file", os.path.join(ROOT, "configs", "config_ves_seg-S_GAN.yml")] + ov)
    rows = open(os.path.join(run2, "metrics.csv")).read().splitlines()
    assert rows[0].startswith("epoch,train_DiceBCELoss,val_DiceBCELoss,Train_DSC") and len(rows) == 2
    vals = [float(v) for v in rows[1].split(",")]
    assert np.isfinite(vals).all() and 0 < vals[1] < 2
    assert "latest_model_model.pth" in os.listdir(os.path.join(run2, "checkpoints"))
    # the loader itself: batched generator pass == per-sample chain (same seeds, same permutation)
    cfg["Train"]["data"] = {"image": {"files": csvs}, "label": {"files": csvs}, "background": {"files": os.path.join(data, "background", "*.png")}}
    cfg["General"].update(seed=9)
    got = {}
    for batched in (True, False):
        torch.manual_seed(6); random.seed(6); np.random.seed(6)
        loader = get_dataset(cfg, "Train", num_workers=0)
        assert loader.fused is None and loader.dataset.transform.batchable()
        if not batched:
            loader.dataset.get_batch = lambda idx, ds=loader.dataset: [ds[j] for j in idx]
        got[batched] = (list(loader), random.random(), float(np.random.random_sample()), float(torch.rand(())))
    assert got[True][1:] == got[False][1:]
    for a, b in zip(got[True][0], got[False][0]):
        assert a["image_path"] == b["image_path"]
        assert a["image"].shape == (4, 1, 1216, 1216) and a["image"].dtype == torch.bfloat16 and a["label"].shape == (4, 1, 1216, 1216)
        assert (a["image"].float() - b["image"].float()).abs().max().item() <= 2e-2
        assert torch.equal(a["label"], b["label"]) and set(a["label"].unique().tolist()) <= {0.0, 1.0}
        assert 0.0 <= a["image"].float().min().item() and a["image"].float().max().item() <= 1.0 + 1e-2
