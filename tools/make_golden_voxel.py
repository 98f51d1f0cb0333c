"""tools/make_golden_voxel.py -- generates tests/golden/voxel_golden.npz by running the reference's own
voxelize_forest (tree2img.py:176-280) in the build container on small volumes (the full
1216x1216x16 volume is 626 MB as float64). Fixture = inputs + outputs only."""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
from vessel_graph_generation.tree2img import voxelize_forest  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "..", "tests", "golden", "voxel_golden.npz")


def main():
    g = np.load(os.path.join(HERE, "..", "tests", "golden", "raster_golden.npz"))
    e = g["graph0_edges"]
    cases = [(e[:600], [152, 152, 4], False, 0, 1), (e[:300], [96, 96, 16], False, 0, 1), (e[:300], [120, 80, 6], True, 0, 1),
             (e[::20], [304, 304, 4], False, 0.002, 0.008)]
    out = {}
    for k, (ed, dims, iz, mn, mx) in enumerate(cases):
        forest = [{"node1": ed[i, 0:3].copy(), "node2": ed[i, 3:6].copy(), "radius": ed[i, 6]} for i in range(len(ed))]
        rl = []
        vol, _ = voxelize_forest(forest, dims, rl, min_radius=mn, max_radius=mx, ignore_z=iz)
        out[f"case{k}_edges"] = ed
        out[f"case{k}_dims"] = np.array(dims)
        out[f"case{k}_args"] = np.array([float(iz), mn, mx])
        out[f"case{k}_vol"] = vol.astype(np.uint8)
        out[f"case{k}_n_radius"] = np.array(len(rl))
        print(k, dims, vol.shape, int((vol > 0).sum()))
    out["n_cases"] = np.array(len(cases))
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
