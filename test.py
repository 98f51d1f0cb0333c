"""Drop-in for the reference's test.py on MI355X (reference test.py:1-90): `--config_file --epoch --num_samples --num_workers`
+ dotted overrides; loads the network named by `General.inference` from `<Output.save_dir>/checkpoints/<epoch>_<net>_model.pth`,
runs every test image through it and its `post_processing.prediction` chain (sigmoid -> 0.5 threshold -> RemoveSmallObjects on
the GPU, csrc/postproc.hip) and writes `<Test.save_dir or Output.save_dir/test>/<mode>_<image name>.png` (uint8(pred * 255),
visualizer.py:330-339) with the native PNG encoder."""
import argparse
import json
import os

import torch
import yaml


def main(argv=None):
    parser = argparse.ArgumentParser(description="")
    parser.add_argument("--config_file", type=str, required=True)
    parser.add_argument("--epoch", type=str, default="best")
    parser.add_argument("--num_samples", type=int, default=9999999)
    parser.add_argument("--num_workers", type=int, default=None)
    args, unknown = parser.parse_known_args(argv)
    assert args.num_samples > 0
    path = os.path.abspath(args.config_file)
    assert os.path.isfile(path), f"Your provided config path {args.config_file} does not exist!"
    with open(path, "r") as stream:
        config = json.load(stream) if path.endswith(".json") else yaml.safe_load(stream)

    from octa_autosegmentation_amd.data.image_dataset import get_dataset, get_post_transformation
    from octa_autosegmentation_amd.models.model import define_model
    from octa_autosegmentation_amd.models.networks import init_weights
    from octa_autosegmentation_amd.utils.config_overrides import apply_cli_overrides_from_unknown_args
    from octa_autosegmentation_amd.utils.enums import Phase
    from octa_autosegmentation_amd.utils.visualizer import plot_sample, plot_single_image
    from train import set_determinism
    apply_cli_overrides_from_unknown_args(config, unknown)
    if config["General"].get("seed") is not None:
        set_determinism(seed=config["General"]["seed"])
    save_dir = config[Phase.TEST].get("save_dir") or config["Output"]["save_dir"] + "/test"
    os.makedirs(save_dir, exist_ok=True)
    device = torch.device(config["General"].get("device") or "cpu")
    print(f"Using device: {device}")
    if device.type == "cuda":
        torch.cuda.set_device(device)
    scaler = torch.amp.GradScaler("cuda", enabled=False)

    test_loader = get_dataset(config, Phase.TEST, num_workers=args.num_workers)
    post_transformations_test = get_post_transformation(config, Phase.TEST)
    input_key = None          # first non-path key of the TRANSFORMED mini-batch, as in the reference (test.py:63-65): keys a transform
                              # deletes (`background`) do not count

    model = define_model(config, phase=Phase.TEST)
    model.initialize_model_and_optimizer(None, init_weights, config, args, scaler, phase=Phase.TEST)
    model.eval()
    # The reference's test.py evaluates in fp32 (no autocast, test.py:79), and so does this one for every network: DynUNet's and -- since
    # round 6 -- the contrast-adaptation generator's (`General.inference: G`) convolutions run the exact-fp32 MFMA kernels
    # (csrc/conv_f32.hip; within 1e-4 of the CPU modules, tests/test_conv_f32_gpu.py, tests/test_networks_golden.py).
    import contextlib
    precision = contextlib.nullcontext
    written = []
    with torch.no_grad():
        for num_sample, test_mini_batch in enumerate(test_loader):
            if num_sample >= args.num_samples:
                break
            if input_key is None:
                input_key = [k for k in test_mini_batch.keys() if not k.endswith("_path")][0]
            test_mini_batch["image"] = test_mini_batch.pop(input_key)
            with precision():
                outputs, _ = model.inference(test_mini_batch, post_transformations_test, device=device, phase=Phase.TEST)
            inference_mode = config["General"].get("inference") or "pred"
            image_name: str = test_mini_batch[f"{input_key}_path"][0].split("/")[-1]
            written.append(plot_single_image(save_dir, outputs["prediction"][0], inference_mode + "_" + image_name))
            if config["Output"].get("save_comparisons"):
                plot_sample(save_dir, test_mini_batch["image"][0], outputs["prediction"][0], None,
                            test_mini_batch[f"{input_key}_path"][0], suffix=f"{inference_mode}_{image_name}", full_size=True)
    test_loader.close()         # one pass: stop the loader's thread
    print(f"wrote {len(written)} predictions to {save_dir}")
    return written


if __name__ == "__main__":
    main()
