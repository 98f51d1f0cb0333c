"""tools/make_golden_sim_masks.py -- generates tests/golden/sim_masks_golden.npz: the imported reference
(tools/make_golden_sim.run_reference) on sampling geometries OTHER than the shipped [76, 76, 1] slab
(simulation_space.py:29-34, 70-76: extent = mask shape / its largest dimension, sinks from the valid voxels, stumps from a random
valid voxel of face 0 along the wall's axis -- for every wall: `np.take(geometry, shape[axis] - 1, axis)` truncates its float index
to 0 -- incl. the z walls, which the reference can only serve with a fixed geometry). Build container only; short runs."""
import copy
import os
import sys

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import make_golden_sim as mg  # noqa: E402  (imports the reference)

OUT = os.path.join(ROOT, "tests", "golden", "sim_masks_golden.npz")


def masks():
    out = {}
    # a thick slab, narrower in x, with a round hole and a blocked corner: three voxel layers, z walls usable
    i, j, k = np.meshgrid(np.arange(48), np.arange(64), np.arange(3), indexing="ij")
    m = np.ones((48, 64, 3), np.uint8)
    m[(i - 20) ** 2 + (j - 30) ** 2 < 49] = 0
    m[:6, :9, :] = 0
    m[40:, 50:, 1:] = 0
    out["thick"] = m
    # a one-voxel slab that is not square and not 76 wide
    i, j = np.meshgrid(np.arange(60), np.arange(40), indexing="ij")
    m2 = np.ones((60, 40, 1), np.uint8)
    m2[(i - 30) ** 2 + (j - 20) ** 2 < 36] = 0
    out["flat"] = m2
    return out


def main():
    base = yaml.safe_load(open(mg.CONFIG))
    g = {"config_yaml": np.array(yaml.safe_dump(base))}
    names = []
    tmp = os.path.join("/tmp", "octa_masks")
    os.makedirs(tmp, exist_ok=True)
    cases = [("thick", dict(x0=True, x1=True, y0=True, y1=True, z0=True, z1=True), 0, 20, 12),
             ("thick", dict(x0=True, x1=False, y0=False, y1=True, z0=True, z1=False), 5, 14, 6),
             ("flat", dict(x0=True, x1=True, y0=True, y1=True, z0=False, z1=False), 2, 20, 12),
             # the walls are drawn by position in the mapping (forest.py:81-91): a mapping that is not in x0 .. z1 order
             ("thick", dict(z1=True, y0=True, x1=True, x0=False, y1=False, z0=True), 9, 14, 6)]
    for mname, m in masks().items():
        g[f"mask_{mname}"] = m
        np.save(os.path.join(tmp, mname + ".npy"), m)
    for mname, walls, seed, i1, i2 in cases:
        cfg = copy.deepcopy(base)
        cfg["Greenhouse"]["SimulationSpace"]["oxygen_sample_geometry_path"] = os.path.join(tmp, mname + ".npy")
        cfg["Forest"]["source_walls"] = walls
        cfg["Greenhouse"]["modes"][0]["I"] = i1
        cfg["Greenhouse"]["modes"][1]["I"] = i2
        r = mg.run_reference(cfg, seed)
        name = f"{mname}_s{seed}_{i1}_{i2}"
        names.append(name)
        g[name + "_seed_I"] = np.array([seed, i1, i2])
        g[name + "_walls"] = np.array([int(bool(walls[k])) for k in ("x0", "x1", "y0", "y1", "z0", "z1")])
        g[name + "_wall_order"] = np.array([("x0", "x1", "y0", "y1", "z0", "z1").index(k) for k in walls])   # order of the mapping
        g[name + "_mask"] = np.array(mname)
        g[name + "_trace"] = r["trace"]
        g[name + "_faz"] = np.array(r["faz"])
        g[name + "_n_art"] = np.array(r["n_art"])
        g[name + "_next"] = np.array([r["next_py"], r["next_np"]])
        g[name + "_csv"] = np.frombuffer(r["csv"].encode(), dtype=np.uint8)
        g[name + "_oxy"] = r["oxy"]
        g[name + "_co2"] = r["co2"]
        print(name, "rows", r["csv"].count("\n") - 1, "oxy", len(r["oxy"]), "co2", len(r["co2"]), flush=True)
    g["names"] = np.array(names)
    np.savez_compressed(OUT, **g)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
