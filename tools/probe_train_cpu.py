"""Where train.py loses time against the bare step: CPU enqueue time of a DynUNet-S step, loader-only rate, loop variants."""
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import yaml  # noqa: E402

import bench  # noqa: E402

ROOT = bench.ROOT


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from octa_autosegmentation_amd.models.segmentation_trainer import SegmentationTrainer
    cfg = {"General": {"amp": True, "model": {"name": "DynUNet", "spatial_dims": 2, "in_channels": 1, "out_channels": 1, "kernel_size": [3, 3, 3, 3, 3],
                                               "strides": [1, 2, 2, 2, 1], "upsample_kernel_size": [1, 2, 2, 2, 1]}},
           "Train": {"lr": 1e-4, "loss": "DiceBCELoss", "epochs": 30, "epochs_decay": 10}}
    torch.manual_seed(0)
    tr = SegmentationTrainer(cfg, dev)
    x = torch.rand(4, 1, 1216, 1216, device=dev)
    y = (torch.rand(4, 1, 1216, 1216, device=dev) > 0.8).float()
    for _ in range(4):
        tr.perform_training_step({"image": x, "label": y})
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(20):
        tr.perform_training_step({"image": x, "label": y})
    t_enq = time.time() - t0
    torch.cuda.synchronize()
    t_all = time.time() - t0
    print(f"bare step: CPU enqueue {t_enq / 20 * 1e3:.2f} ms/step, GPU-complete {t_all / 20 * 1e3:.2f} ms/step", flush=True)
    # loader alone
    import generate_vessel_graph
    from octa_autosegmentation_amd.data.image_dataset import get_dataset
    from octa_autosegmentation_amd.utils import configs
    tmp = tempfile.mkdtemp(prefix="octa_probe_", dir="/dev/shm")
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        generate_vessel_graph.main(["--config_file", configs.GENERATOR_CONFIG, "--num_samples", "64", "--seed", "1", "--output.directory", os.path.join(tmp, "graphs")])
    c = yaml.safe_load(open(os.path.join(ROOT, "configs", "config_ves_seg-S.yml")))
    csvs = os.path.join(tmp, "graphs", "**", "*.csv")
    c["Train"]["data"] = {"image": {"files": csvs}, "label": {"files": csvs}}
    c["General"]["seed"] = 3
    for workers in (0, 1):
        loader = get_dataset(c, "Train", num_workers=workers)
        for ep in range(3):
            torch.cuda.synchronize()
            t0 = time.time()
            n = 0
            for b in loader:
                n += 1
            t_cpu = time.time() - t0
            torch.cuda.synchronize()
            print(f"loader only (workers={workers}) epoch {ep}: {n} batches, {t_cpu / n * 1e3:.2f} ms/batch consumer-side, {(time.time() - t0) / n * 1e3:.2f} ms/batch complete", flush=True)
        loader.close()
    # loop variants with the loader
    loader = get_dataset(c, "Train", num_workers=1)
    for variant in ("step only", "step + metrics"):
        from octa_autosegmentation_amd.utils.metrics import MetricsManager
        from octa_autosegmentation_amd.data.image_dataset import get_post_transformation
        metrics = MetricsManager()
        post = get_post_transformation(c, "Train")
        for ep in range(3):
            torch.cuda.synchronize()
            t0 = time.time()
            n = 0
            for b in loader:
                if variant == "step only":
                    tr.perform_training_step(b)
                else:
                    out, losses = tr.impl.perform_training_step(b, None, post, dev)
                    with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
                        tr.impl.compute_metric(out, metrics)
                n += 1
            torch.cuda.synchronize()
            print(f"{variant} epoch {ep}: {4 * n / (time.time() - t0):.1f} imgs/s", flush=True)
    loader.close()


if __name__ == "__main__":
    main()
