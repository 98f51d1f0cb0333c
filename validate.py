"""Drop-in for the reference's validate.py on MI355X (reference validate.py:1-70): loads the `best` (or `--epoch`) checkpoint
of the network named by `General.inference`, runs the validation set with batch size 1 and prints the aggregated metrics
rounded to three decimals."""
import argparse
import json
import os

import torch
import yaml


def main(argv=None):
    parser = argparse.ArgumentParser(description="")
    parser.add_argument("--config_file", type=str, required=True)
    parser.add_argument("--epoch", type=str, default="best")
    parser.add_argument("--num_workers", type=int, default=None)
    args, unknown = parser.parse_known_args(argv)
    path = os.path.abspath(args.config_file)
    assert os.path.isfile(path), f"Your provided config path {args.config_file} does not exist!"
    with open(path, "r") as stream:
        config = json.load(stream) if path.endswith(".json") else yaml.safe_load(stream)

    from octa_autosegmentation_amd.data.image_dataset import get_dataset, get_post_transformation
    from octa_autosegmentation_amd.models.model import define_model
    from octa_autosegmentation_amd.models.networks import init_weights
    from octa_autosegmentation_amd.utils.config_overrides import apply_cli_overrides_from_unknown_args
    from octa_autosegmentation_amd.utils.enums import Phase
    from octa_autosegmentation_amd.utils.metrics import MetricsManager
    apply_cli_overrides_from_unknown_args(config, unknown)
    config[Phase.VALIDATION]["batch_size"] = 1
    val_loader = get_dataset(config, Phase.VALIDATION, num_workers=args.num_workers)
    post_transformations_val = get_post_transformation(config, phase=Phase.VALIDATION)
    device = torch.device(config["General"].get("device") or "cpu")
    if device.type == "cuda":
        torch.cuda.set_device(device)
    scaler = torch.amp.GradScaler("cuda", enabled=False)
    model = define_model(config, phase=Phase.VALIDATION)
    model.initialize_model_and_optimizer(None, init_weights, config, args, scaler, phase=Phase.VALIDATION)
    metrics = MetricsManager(Phase.VALIDATION)
    model.eval()
    with torch.no_grad():
        for val_mini_batch in val_loader:
            # fp32 like the reference's validate.py (no autocast): DynUNet's convolutions on the exact-fp32 MFMA kernel (csrc/conv_f32.hip)
            outputs, losses = model.inference(val_mini_batch, post_transformations_val, device=device, phase=Phase.VALIDATION)
            model.compute_metric(outputs, metrics)
    val_loader.close()          # one pass: the loader's thread must not go on preparing an epoch nobody reads (nor touch HIP at interpreter shutdown)
    result = {k: float(str(round(v, 3))) for k, v in metrics.aggregate_and_reset(Phase.VALIDATION).items()}
    print(f"Metrics: {result}")
    return result


if __name__ == "__main__":
    main()
