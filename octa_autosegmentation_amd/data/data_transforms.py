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


def get_data_augmentations(aug_config, dtype=torch.float32, seed=None):
    """Subset of the reference's registry (data_transforms.py:587-611): names outside the hot path raise."""
    table = {"LoadGraphAndFilterByRandomRadiusd": LoadGraphAndFilterByRandomRadiusd}
    out = []
    for d in aug_config:
        d = dict(d)
        name = d.pop("name")
        if name not in table:
            raise NotImplementedError(f"transform {name} is outside the MI355X hot path (MONAI is not a dependency)")
        out.append(table[name](**d))
    return out
