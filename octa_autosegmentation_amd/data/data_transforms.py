"""Graph -> tensor transform of the training loader (reference data/data_transforms.py:358-387
LoadGraphAndFilterByRandomRadiusd), without MONAI: the dict-in / dict-out call convention and the
constructor arguments are kept; every key is rasterised on the GPU at its own resolution / radius
threshold with the blackdict of the first key shared by the later ones (consistent dropout)."""
import csv
import pickle

import numpy as np
import torch

from ..vessel_graph_generation.tree2img import rasterize_forest


class LoadGraphAndFilterByRandomRadiusd:
    def __init__(self, keys, allow_missing_keys: bool = False, image_resolutions=[[304, 304]], min_radius=[0],
                 max_dropout_prob=0, MIP_axis=2) -> None:
        self.keys = [keys] if isinstance(keys, str) else list(keys)
        self.allow_missing_keys = allow_missing_keys
        self.min_radius = min_radius
        self.image_resolutions = image_resolutions
        self.max_dropout_prob = max_dropout_prob
        self.MIP_axis = MIP_axis

    def __call__(self, data):
        data = dict(data)
        if "blackdict" in data:
            with open(data["blackdict"], mode="rb") as file:
                blackdict = pickle.load(file)
        else:
            blackdict = None
        for i, key in enumerate(self.keys):
            if key not in data and self.allow_missing_keys:
                continue
            with open(data[key], newline='') as csvfile:
                f = list(csv.DictReader(csvfile))
            img, blackdict = rasterize_forest(f, self.image_resolutions[i], self.MIP_axis, min_radius=self.min_radius[i],
                                              max_dropout_prob=self.max_dropout_prob, blackdict=blackdict)
            data[key] = torch.tensor(img.astype(np.float32))
        return data


class SpeckleBrightnesd:
    """Speckle component of the noise model (reference data_transforms.py:25-42): a 9x9 random control grid in
    [0.5, 1) bilinearly upsampled to the image, R = C - U*(1-C), img*R, then renormalised exactly like the
    reference (divide by the max, subtract the min). Runs on whatever device the tensor lives on."""

    def __init__(self, keys, allow_missing_keys: bool = False) -> None:
        self.keys = [keys] if isinstance(keys, str) else list(keys)
        self.allow_missing_keys = allow_missing_keys

    def __call__(self, data):
        data = dict(data)
        for key in self.keys:
            if key not in data and self.allow_missing_keys:
                continue
            img = data[key]
            c = torch.rand((1, 1, 9, 9), device=img.device) * 0.5 + 0.5
            C = torch.nn.functional.interpolate(c, size=img.shape[-2:], mode="bilinear").squeeze(0)
            R = C - (torch.rand_like(C) * (1 - C))
            img = img * R
            img = img / img.max()
            img = img - img.min()
            data[key] = img
        return data


class AddRandomBackgroundNoised:
    """max(img, background * U(0,1)) with uniform noise when no background tile is given (reference :498-516)."""

    def __init__(self, keys, delete_background=True) -> None:
        self.keys = [keys] if isinstance(keys, str) else list(keys)
        self.delete_background = delete_background

    def __call__(self, data):
        data = dict(data)
        for key in self.keys:
            if key in data:
                img = data[key]
                noise = data["background"].to(img.device) if "background" in data else torch.rand_like(img)
                speckle = torch.from_numpy(np.random.uniform(0, 1, tuple(img.shape))).to(img.device, img.dtype)
                data[key] = torch.maximum(img, noise * speckle)
        if self.delete_background and "background" in data:
            del data["background"]
        return data


class ImageToImageTranslationd:
    """Frozen-generator contrast adaptation (reference :327-356) -- on the GPU instead of inside CPU loader
    workers (6.1 s per image on the CPU). `model` may be passed directly; otherwise
    resnetGenerator9 weights are loaded from `model_path` (checkpoint dict with key 'model')."""

    def __init__(self, model_path=None, keys=("image",), model_config: dict = None, allow_missing_keys: bool = False,
                 model=None, device=None) -> None:
        from ..models.networks import MODEL_DICT
        self.keys = [keys] if isinstance(keys, str) else list(keys)
        self.allow_missing_keys = allow_missing_keys
        if model is None:
            if model_config is not None and model_config.get("name", "resnetGenerator9") != "resnetGenerator9":
                raise NotImplementedError("only resnetGenerator9 translation models are on the MI355X hot path")
            model = MODEL_DICT["resnetGenerator9"]()
            if model_path is not None:
                ckpt = torch.load(model_path, map_location="cpu")
                model.load_state_dict(ckpt["model"])
        self.device = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.model = model.to(self.device).eval()

    def __call__(self, data):
        data = dict(data)
        for key in self.keys:
            if key not in data and self.allow_missing_keys:
                continue
            with torch.no_grad():
                img = data[key]
                data[key] = self.model(img.float().unsqueeze(0).to(self.device)).squeeze(0)
        return data


def get_data_augmentations(aug_config, dtype=torch.float32, seed=None):
    """Subset of the reference's registry (data_transforms.py:587-611): names outside the hot path raise."""
    table = {"LoadGraphAndFilterByRandomRadiusd": LoadGraphAndFilterByRandomRadiusd, "SpeckleBrightnesd": SpeckleBrightnesd,
             "AddRandomBackgroundNoised": AddRandomBackgroundNoised, "ImageToImageTranslationd": ImageToImageTranslationd}
    out = []
    for d in aug_config:
        d = dict(d)
        name = d.pop("name")
        if name not in table:
            raise NotImplementedError(f"transform {name} is outside the MI355X hot path (MONAI is not a dependency)")
        out.append(table[name](**d))
    return out
