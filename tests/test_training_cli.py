"""CPU: the training-side boundary -- ModelInterface / define_model / train.py / test.py / validate.py, the transform registry, the
loader and the run-directory formats -- on a small PNG dataset (the graph loader and RemoveSmallObjects need the GPU: they are
covered by tests/test_training_cli_gpu.py). Reference: train.py:29-230, test.py, validate.py, models/model_interface_abc.py,
data/image_dataset.py, utils/visualizer.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _png_dataset(root, n=6, size=48):
    from PIL import Image
    rng = np.random.default_rng(0)
    os.makedirs(os.path.join(root, "images")); os.makedirs(os.path.join(root, "labels"))
    for i in range(n):
        lab = np.zeros((size, size), np.uint8)
        for _ in range(3):
            r, c = rng.integers(4, size - 4, 2)
            lab[r - 2:r + 2, :] = 255; lab[:, c - 1:c + 1] = 255
        img = np.clip(lab * 0.6 + rng.normal(40, 12, lab.shape), 0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(root, "images", f"s{i}.png"))
        Image.fromarray(lab).save(os.path.join(root, "labels", f"s{i}.png"))


def _config(root, out, device="cpu"):
    aug = lambda keys, rand: ([{"name": "LoadImaged", "keys": keys, "image_only": True}, {"name": "ToGrayScaled", "keys": keys},
                               {"name": "ScaleIntensityd", "keys": keys, "minv": 0, "maxv": 1},
                               {"name": "EnsureChannelFirstd", "keys": keys, "strict_check": False, "channel_dim": "no_channel"},
                               {"name": "Resized", "keys": keys, "spatial_size": [64, 64], "mode": "bilinear"},
                               {"name": "Rotate90d", "keys": keys, "k": 1}, {"name": "Flipd", "keys": keys, "spatial_axis": 0}]
                              + ([{"name": "RandFlipd", "keys": keys, "prob": 0.5, "spatial_axis": [0, 1]},
                                  {"name": "RandRotate90d", "keys": keys, "prob": 0.75},
                                  {"name": "RandRotated", "keys": keys, "prob": 1, "range_x": 0.17, "padding_mode": "zeros"}] if rand else [])
                              + ([{"name": "AsDiscreted", "keys": ["label"], "threshold": 0.5}] if "label" in keys else [])
                              + [{"name": "CastToTyped", "keys": keys, "dtype": "dtype"}])
    post = {"prediction": [{"name": "Activations", "sigmoid": True}, {"name": "AsDiscrete", "threshold": 0.5}],
            "label": [{"name": "CastToType", "dtype": "uint8"}]}
    data = {"image": {"files": os.path.join(root, "images", "*.png")}, "label": {"files": os.path.join(root, "labels", "*.png")}}
    return {"General": {"amp": False, "device": device, "task": "ves-seg", "seed": 5,
                        "model": {"name": "DynUNet", "spatial_dims": 2, "in_channels": 1, "out_channels": 1, "kernel_size": [3, 3, 3],
                                  "strides": [1, 2, 1], "upsample_kernel_size": [2, 1], "filters": [4, 8, 8]}},
            "Train": {"data": data, "epochs": 2, "epochs_decay": 1, "val_interval": 1, "save_interval": 2, "batch_size": 2, "lr": 1e-3,
                      "loss": "DiceBCELoss", "AT": False, "data_augmentation": aug(["image", "label"], True), "post_processing": post},
            "Validation": {"batch_size": 2, "data": data, "data_augmentation": aug(["image", "label"], False), "post_processing": post},
            "Test": {"batch_size": 1, "data": {"image": data["image"]}, "save_comparisons": False,
                     "data_augmentation": aug(["image"], False), "post_processing": post},
            "Output": {"save_dir": out, "save_to_disk": True, "save_to_tensorboard": False}}


def test_phase_enum_is_a_config_key():
    from octa_autosegmentation_amd.utils.enums import Phase, Task
    cfg = {"Train": 1, "Validation": 2}
    assert cfg[Phase.TRAIN] == 1 and Phase.VALIDATION in cfg and Phase.TEST not in cfg
    assert Phase.TRAIN == "Train" and str(Phase.TEST) == "Test" and Task.GAN_VESSEL_SEGMENTATION == "gan-ves-seg"
    assert [p for p in Phase if p in cfg] == [Phase.TRAIN, Phase.VALIDATION]


def test_load_rotate_flip_chain_restores_the_orientation(tmp_path):
    """LoadImaged hands over [W, H] like MONAI's PILReader; the configs' Rotate90d(k=1) + Flipd(spatial_axis=0) undo it."""
    from PIL import Image
    from octa_autosegmentation_amd.data.data_transforms import Compose, get_data_augmentations
    a = (np.arange(30 * 20).reshape(30, 20) % 251).astype(np.uint8)
    p = str(tmp_path / "a.png")
    Image.fromarray(a).save(p)
    chain = Compose(get_data_augmentations([{"name": "LoadImaged", "keys": ["image"], "image_only": True}, {"name": "ToGrayScaled", "keys": ["image"]},
                                            {"name": "EnsureChannelFirstd", "keys": ["image"], "channel_dim": "no_channel"},
                                            {"name": "Rotate90d", "keys": ["image"], "k": 1}, {"name": "Flipd", "keys": ["image"], "spatial_axis": 0}], seed=1))
    loaded = Compose(get_data_augmentations([{"name": "LoadImaged", "keys": ["image"]}], seed=1))({"image": p})["image"]
    assert tuple(loaded.shape) == (20, 30)
    out = chain({"image": p})["image"].cpu().numpy()
    assert out.shape == (1, 30, 20) and (out[0] == a).all()
    rgb = np.stack([a, a // 2, 255 - a], axis=-1)
    Image.fromarray(rgb).save(p)
    grey = chain({"image": p})["image"].cpu().numpy()[0]
    assert (grey == np.array(Image.fromarray(rgb).convert("L"))).all()


def test_transforms_follow_the_torch_ops_monai_delegates_to():
    import torch.nn.functional as F
    from octa_autosegmentation_amd.data import data_transforms as T
    g = torch.Generator().manual_seed(0)
    x = torch.rand(1, 40, 40, generator=g) * 7 + 3
    y = T.ScaleIntensityd(["a"], minv=0, maxv=1)({"a": x})["a"]
    assert torch.allclose(y, (x - x.min()) / (x.max() - x.min())) and float(y.min()) == 0 and abs(float(y.max()) - 1) < 1e-6
    assert torch.equal(T.ScaleIntensityd(["a"], minv=0.5, maxv=1)({"a": torch.full((1, 4, 4), 3.0)})["a"], torch.full((1, 4, 4), 1.5))
    r = T.Resized(["a"], [64, 96], mode="bilinear")({"a": x})["a"]
    assert torch.equal(r, F.interpolate(x[None], size=(64, 96), mode="bilinear").squeeze(0))
    assert torch.equal(T.Flipd(["a"], spatial_axis=[0, 1])({"a": x})["a"], torch.flip(x, (1, 2)))
    assert torch.equal(T.Rotate90d(["a"], k=3)({"a": x})["a"], torch.rot90(x, 3, (1, 2)))
    assert torch.equal(T.AsDiscreted(["a"], threshold=5.0)({"a": x})["a"], (x >= 5.0).float())
    # random transforms: same seed, same decisions; image and label of a sample share them
    cfg = [{"name": "RandFlipd", "keys": ["a", "b"], "prob": 0.5, "spatial_axis": [0, 1]}, {"name": "RandRotate90d", "keys": ["a", "b"], "prob": 0.75},
           {"name": "RandRotated", "keys": ["a", "b"], "prob": 1, "range_x": 0.17, "padding_mode": "zeros"}]
    lab = (x > 6).float()
    outs = []
    for _ in range(2):
        chain = T.Compose(T.get_data_augmentations(cfg, seed=11))
        outs.append([chain({"a": x, "b": lab}) for _ in range(4)])
    for s0, s1 in zip(*outs):
        assert torch.equal(s0["a"], s1["a"]) and torch.equal(s0["b"], s1["b"])
    assert any(not torch.equal(outs[0][0]["a"], o["a"]) for o in outs[0][1:])
    # RandRotated = rotation about the centre, bilinear, zeros outside (F.affine_grid + F.grid_sample)
    rot = T.RandRotated(["a"], range_x=0.17, prob=1.0, padding_mode="zeros").set_random_state(3)
    got = rot({"a": x})["a"]
    ang = np.random.RandomState(3); ang.rand(); ang = ang.uniform(-0.17, 0.17)
    c, s = float(np.cos(ang)), float(np.sin(ang))
    grid = F.affine_grid(torch.tensor([[[c, -s, 0.0], [s, c, 0.0]]]), (1, 1, 40, 40), align_corners=False)
    assert torch.allclose(got, F.grid_sample(x[None], grid, mode="bilinear", padding_mode="zeros", align_corners=False)[0], atol=1e-6)
    # CastToType* placeholders
    t = T.get_data_augmentations([{"name": "CastToTyped", "keys": ["a"], "dtype": "dtype"}], seed=0, dtype=torch.bfloat16)[0]
    assert t({"a": x})["a"].dtype == torch.bfloat16
    assert T.get_data_augmentations([{"name": "CastToType", "dtype": "uint8"}], seed=0)[0](x).dtype == torch.uint8
    with pytest.raises(NotImplementedError):
        T.get_data_augmentations([{"name": "RandGaussianNoised", "keys": ["a"]}], seed=0)


def test_unaligned_zip_dataset_draws_like_the_reference():
    import random
    from octa_autosegmentation_amd.data.unalignedZipDataset import UnalignedZipDataset
    data = {"real_A": [f"a{i}" for i in range(5)], "real_A_seg": [f"a{i}" for i in range(5)], "real_B": [f"b{i}" for i in range(7)],
            "background": [f"g{i}" for i in range(3)]}
    ds = UnalignedZipDataset(data, lambda d: d)
    assert len(ds) == 7
    random.seed(4)
    got = [ds[i] for i in range(7)]
    random.seed(4)
    for i, s in enumerate(got):
        assert s["real_A"] == s["real_A_path"] == f"a{i % 5}" and s["real_A_seg"] == f"a{i % 5}"
        assert s["real_B"] == f"b{random.randint(0, 6)}" and s["background"] == f"g{random.randint(0, 2)}"


def test_loader_shards_batches_across_ranks():
    from octa_autosegmentation_amd.data.image_dataset import DeviceLoader, ListDataset
    ds = ListDataset([{"x": torch.tensor([float(i)]), "x_path": f"p{i}"} for i in range(10)], lambda d: d)
    seen = []
    for rank in range(2):
        torch.manual_seed(9)
        ld = DeviceLoader(ds, batch_size=2, shuffle=True, num_workers=0)
        ld.shard = (rank, 2)
        bs = list(ld)
        assert len(bs) == len(ld) == 3 and all(b["x"].shape == (2, 1) and len(b["x_path"]) == 2 for b in bs)
        seen.append([float(v) for b in bs for v in b["x"].flatten()])
    assert sorted(set(seen[0]) | set(seen[1])) == [float(i) for i in range(10)]          # 5 batches over 2 ranks, one repeated to even out
    assert len(set(seen[0]) & set(seen[1])) == 2


def test_metrics_manager_values():
    from octa_autosegmentation_amd.utils.enums import Phase
    from octa_autosegmentation_amd.utils.metrics import MetricsManager
    rng = np.random.default_rng(1)
    y = torch.from_numpy((rng.random((1, 32, 32)) > 0.6).astype(np.float32))
    p = torch.from_numpy((rng.random((1, 32, 32)) > 0.5).astype(np.float32))
    m = MetricsManager(Phase.VALIDATION)
    m([p], [y])
    out = m.aggregate_and_reset(Phase.VALIDATION)
    pb, yb = p.numpy().astype(bool).ravel(), y.numpy().astype(bool).ravel()
    tp, tn, fp, fn = (pb & yb).sum(), (~pb & ~yb).sum(), (pb & ~yb).sum(), (~pb & yb).sum()
    assert abs(out["Validation_DSC"] - 2 * tp / (yb.sum() + pb.sum())) < 1e-6
    assert abs(out["Validation_IoU"] - tp / (tp + fp + fn)) < 1e-6
    assert abs(out["Validation_ACC"] - (tp + tn) / pb.size) < 1e-6
    assert abs(out["Validation_Recall"] - tp / (tp + fn)) < 1e-6 and abs(out["Validation_Precision"] - tp / (tp + fp)) < 1e-6
    tpr, fpr = tp / (tp + fn), fp / (fp + tn)
    assert abs(out["Validation_AUC"] - (0.5 * tpr * fpr + (1 - fpr) * (tpr + 1) / 2)) < 1e-6     # two-level ROC curve, trapezoid
    assert MetricsManager(Phase.TRAIN).get_comp_metric(Phase.VALIDATION) == "Validation_DSC"
    assert set(MetricsManager(Phase.TRAIN).metrics) == {"DSC", "IoU"}


def test_train_test_validate_clis_and_resume(tmp_path):
    """train.py for two epochs, test.py and validate.py on its `best` checkpoint, then a resumed run: file names, dict keys,
    metrics.csv layout and the cloning of a run directory follow the reference (train.py:165-194, utils/visualizer.py)."""
    import test as test_cli
    import train as train_cli
    import validate as validate_cli
    from PIL import Image
    root, out = str(tmp_path / "data"), str(tmp_path / "results")
    _png_dataset(root)
    cfg_path = str(tmp_path / "cfg.yml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(_config(root, out), f)
    run = train_cli.main(["--config_file", cfg_path, "--num_workers", "0"])
    assert os.path.dirname(run) == out and os.path.isfile(os.path.join(run, "config.yml")) and os.path.isfile(os.path.join(run, "architecture.txt"))
    rows = open(os.path.join(run, "metrics.csv")).read().splitlines()
    assert rows[0] == "epoch,train_DiceBCELoss,val_DiceBCELoss,Train_DSC,Train_IoU,Validation_DSC,Validation_IoU" and len(rows) == 3
    assert rows[1].startswith("0,") and rows[2].startswith("1,")
    names = set(os.listdir(os.path.join(run, "checkpoints")))
    assert {f"{t}_{n}_model.pth" for t in ("latest", "2", "best") for n in ("optimizer", "model")} <= names
    ck = torch.load(os.path.join(run, "checkpoints", "latest_model_model.pth"), weights_only=False)
    assert set(ck) == {"epoch", "model", "optimizer", "config"} and ck["epoch"] == 2 and ck["optimizer"] is None
    assert any(n.startswith("sample_train_latest") or n == "sample_train_latest.png" for n in os.listdir(run))
    run_cfg = os.path.join(run, "config.yml")
    # inference CLIs against the run directory's own config (Output.save_dir points at the run)
    written = test_cli.main(["--config_file", run_cfg, "--epoch", "best", "--num_workers", "0"])
    assert len(written) == 6 and all(os.path.basename(w).startswith("model_s") for w in written)   # General.inference resolves to "model" (base_model_abc.py:98)
    pred = np.array(Image.open(written[0]))
    assert pred.shape == (64, 64) and set(np.unique(pred)) <= {0, 255}
    res = validate_cli.main(["--config_file", run_cfg, "--epoch", "best", "--num_workers", "0"])
    assert {"Validation_DSC", "Validation_IoU", "Validation_AUC", "Validation_ACC"} <= set(res) and 0 <= res["Validation_DSC"] <= 1
    # resume: weights and optimiser state of epoch 2 come back, the log is cloned into a fresh run directory
    import time
    time.sleep(1.1)                                         # run directories are named by the second
    run2 = train_cli.main(["--config_file", run_cfg, "--start_epoch", "2", "--epoch", "latest", "--Train.epochs", "3", "--num_workers", "0"])
    assert run2 != run and os.path.dirname(run2) == out
    rows2 = open(os.path.join(run2, "metrics.csv")).read().splitlines()
    assert rows2[:3] == rows and len(rows2) == 4 and rows2[3].startswith("2,")
    ck2 = torch.load(os.path.join(run2, "checkpoints", "latest_optimizer_model.pth"), weights_only=False)
    assert ck2["epoch"] == 3 and ck2["optimizer"]["state"][0]["step"] == 9          # 3 steps per epoch, 3 epochs in total


def _two_rank_train(rank, world, port, cfg_path, outfile):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import train as train_cli
    run = train_cli.main(["--config_file", cfg_path, "--num_workers", "0"])
    if rank == 0:
        open(outfile, "w").write(run)
    import torch.distributed as dist
    dist.destroy_process_group()


def test_train_cli_two_ranks_gloo(tmp_path):
    """The data-parallel path of train.py on two CPU ranks (gloo): shared permutation, every rank a share of the batches, one run
    directory written by rank 0."""
    import torch.multiprocessing as mp
    root, out = str(tmp_path / "data"), str(tmp_path / "results")
    _png_dataset(root, n=8)
    cfg = _config(root, out)
    cfg["Train"]["epochs"] = 1
    cfg_path = str(tmp_path / "cfg.yml")
    with open(cfg_path, "w") as f:
        yaml.safe_dump(cfg, f)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    outfile = str(tmp_path / "run.txt")
    mp.spawn(_two_rank_train, args=(2, port, cfg_path, outfile), nprocs=2, join=True)
    run = open(outfile).read()
    assert len(os.listdir(out)) == 1 and os.path.isfile(os.path.join(run, "metrics.csv"))
    ck = torch.load(os.path.join(run, "checkpoints", "latest_optimizer_model.pth"), weights_only=False)
    assert ck["optimizer"]["state"][0]["step"] == 2        # 4 batches of 2 over 2 ranks
