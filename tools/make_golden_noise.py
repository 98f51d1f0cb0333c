"""tools/make_golden_noise.py -- generates tests/golden/noise_golden.npz: outputs of the reference's OWN noise transforms
(data/data_transforms.py:25-42 SpeckleBrightnesd, :498-516 AddRandomBackgroundNoised) on seeded inputs (row a17).

Runs ONLY in the build container: imports /root/reference/data/data_transforms.py with its absent dependencies mocked (monai,
skimage, the model files; `monai.transforms.MapTransform` is given MONAI's documented minimal behaviour: `keys` as a tuple and
`allow_missing_keys`). The fixture stores inputs, seeds and outputs; the tests re-seed the same generators (torch.manual_seed,
np.random.seed) and compare this repository's transforms -- CPU tensors bit for bit, the HIP kernels within fp32 rounding."""
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden", "noise_golden.npz")


def image(shape, k):
    n = int(np.prod(shape))
    v = 0.5 + 0.5 * np.sin(np.arange(n, dtype=np.float64) * 0.0137 * (k + 1) + np.arange(n, dtype=np.float64) ** 2 * 1e-7)
    return torch.from_numpy(v.astype(np.float32)).reshape(shape)


def main():
    sys.path.insert(0, "/root/reference")
    monai = MagicMock()

    class MapTransform:
        def __init__(self, keys, allow_missing_keys=False):
            self.keys = (keys,) if isinstance(keys, str) else tuple(keys)
            self.allow_missing_keys = allow_missing_keys

    class Randomizable:
        pass

    class Transform:
        pass

    monai.transforms.MapTransform, monai.transforms.Randomizable, monai.transforms.Transform = MapTransform, Randomizable, Transform
    monai.transforms.__all__ = []
    for m in ["monai", "monai.config", "monai.transforms", "monai.data", "monai.losses", "monai.networks", "monai.networks.nets", "skimage", "skimage.draw",
              "skimage.filters", "skimage.morphology", "nibabel", "prettytable", "natsort", "torchvision", "torchvision.transforms",
              "torchvision.transforms.functional", "models.networks", "models.noise_model", "matplotlib", "matplotlib.pyplot", "matplotlib.figure",
              "matplotlib.collections", "matplotlib.backends", "matplotlib.backends.backend_agg"]:
        sys.modules.setdefault(m, MagicMock())
    sys.modules["monai"] = monai
    sys.modules["monai.transforms"] = monai.transforms
    import importlib
    ref = importlib.import_module("data.data_transforms")
    out = {}
    for k, shape in enumerate(((1, 40, 56), (1, 96, 128))):
        img = image(shape, k + 1)
        torch.manual_seed(100 + k)
        out[f"speckle_{k}_in"] = img.numpy().copy()
        out[f"speckle_{k}_out"] = ref.SpeckleBrightnesd(["image"])({"image": img.clone()})["image"].numpy()
        bg = image(shape, k + 7).flip(-1)
        np.random.seed(200 + k)
        d = ref.AddRandomBackgroundNoised(["image"])({"image": img.clone(), "background": bg.clone()})
        out[f"bg_{k}_noise"] = bg.numpy().copy()
        out[f"bg_{k}_out"] = d["image"].numpy()
        assert "background" not in d
        np.random.seed(300 + k); torch.manual_seed(300 + k)
        out[f"bg_{k}_out_nobg"] = ref.AddRandomBackgroundNoised(["image"])({"image": img.clone()})["image"].numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: (v.shape, v.dtype) for k, v in out.items()})


if __name__ == "__main__":
    main()
