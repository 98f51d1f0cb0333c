"""Shared by the simulator tests: cases of the reference-made fixtures."""
import os

import numpy as np


def mask_cases(tmp_path):
    """tests/golden/sim_masks_golden.npz (tools/make_golden_sim_masks.py): reference runs on a three-voxel-thick, non-square mask with
    all six source walls (forest.py:153-181: the z walls), subsets of them (one in a mapping that is not in x0 .. z1 order: the wall is
    drawn by position, forest.py:81-91), and a flat 60 x 40 mask."""
    import yaml
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sim_masks_golden.npz"))
    wall_names = ("x0", "x1", "y0", "y1", "z0", "z1")
    for name in (str(n) for n in g["names"]):
        cfg = yaml.safe_load(str(g["config_yaml"]))
        seed, i1, i2 = (int(v) for v in g[name + "_seed_I"])
        path = str(tmp_path / (name + ".npy"))
        np.save(path, g["mask_" + str(g[name + "_mask"])])
        cfg["Greenhouse"]["SimulationSpace"]["oxygen_sample_geometry_path"] = path
        cfg["Forest"]["source_walls"] = {wall_names[k]: bool(g[name + "_walls"][k]) for k in g[name + "_wall_order"]}
        cfg["Greenhouse"]["modes"][0]["I"], cfg["Greenhouse"]["modes"][1]["I"] = i1, i2
        yield name, cfg, seed, g
