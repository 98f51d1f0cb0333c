"""tools/make_golden_networks.py -- generates tests/golden/networks_golden.npz.

Runs ONLY in the build container: imports the reference's models/networks.py from /root/reference (read-only). The
modules it imports at file scope but the recorded classes never touch (monai, skimage, nibabel, prettytable, natsort,
torchvision, and the reference's other model files) are absent from the image and are replaced by
unittest.mock.MagicMock entries in sys.modules for the duration of the import, as SURVEY.md 8c describes. Recorded, in
fp32 on the CPU:
  * resnetGenerator9 / patchGAN70x70 with every parameter set by the closed form `fill` below (in state_dict order) on
    a closed-form 1x1x64x64 / 1x1x304x304 input: state_dict keys + shapes, the full outputs at 64^2 and crops +
    float64 sums at 304^2;
  * Downsample(4) / Upsample(4) on a closed-form 1x4x16x16 and an odd 1x4x15x13 input.
Fixtures are data only; tests/test_models.py rebuilds the same weights from the formula.
"""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden", "networks_golden.npz")


def fill(shape, k):
    """Deterministic tensor: sin of an index ramp; k decorrelates tensors. Scaled like a kaiming init (1/sqrt(fan_in))."""
    n = int(np.prod(shape))
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
    v = np.sin(np.arange(n, dtype=np.float64) * (0.37 + 0.011 * k) + 0.5 * k) / np.sqrt(max(fan_in, 1))
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def image(shape, k):
    n = int(np.prod(shape))
    v = 0.5 + 0.5 * np.sin(np.arange(n, dtype=np.float64) * 0.0137 * (k + 1) + np.arange(n, dtype=np.float64) ** 2 * 1e-7)
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def load_formula_weights(net):
    sd = net.state_dict()
    for k, (name, t) in enumerate(sd.items()):
        if name.endswith("filt"):
            continue                      # the fixed binomial filters stay as constructed
        sd[name] = fill(tuple(t.shape), k)
    net.load_state_dict(sd)
    return [(name, tuple(t.shape)) for name, t in sd.items()]


def main():
    sys.path.insert(0, "/root/reference")
    for m in ["monai", "monai.networks", "monai.networks.nets", "monai.networks.blocks", "monai.networks.layers", "skimage", "skimage.filters",
              "nibabel", "prettytable", "natsort", "torchvision", "torchvision.models", "torchvision.transforms",
              "models.gan_seg_model", "models.oof", "models.frangi", "models.skrgan", "models.nice_gan", "models.cycle_gan", "models.cut",
              "models.negcut", "models.dclgan"]:
        sys.modules.setdefault(m, MagicMock())
    import importlib
    ref = importlib.import_module("models.networks")

    out = {}
    torch.manual_seed(0)
    with torch.no_grad():
        for name, ctor, sizes in (("G", ref.resnetGenerator9, (64, 304)), ("D", ref.patchGAN70x70, (64, 304))):
            net = ctor().eval()
            keys = load_formula_weights(net)
            out[f"{name}_keys"] = np.array([k for k, _ in keys])
            out[f"{name}_shapes"] = np.array([",".join(map(str, s)) for _, s in keys])
            for s in sizes:
                y = net(image((1, 1, s, s), 1 if name == "G" else 2)).double().numpy()
                out[f"{name}_sum_{s}"] = np.array([y.sum(), np.abs(y).sum()])
                if s == 64:
                    out[f"{name}_out_{s}"] = y.astype(np.float32)
                else:
                    out[f"{name}_crop_{s}"] = y[0, 0, :24, :24].astype(np.float32)
                    out[f"{name}_crop2_{s}"] = y[0, 0, -24:, -24:].astype(np.float32)
        for shape in ((1, 4, 16, 16), (1, 4, 15, 13)):
            x = image(shape, 3)
            tag = f"{shape[2]}x{shape[3]}"
            out[f"down_{tag}"] = ref.Downsample(4)(x).numpy()
            out[f"up_{tag}"] = ref.Upsample(4)(x).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
