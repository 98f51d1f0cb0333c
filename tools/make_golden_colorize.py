"""Build container: run the REFERENCE's rasterize_forest(colorize=...) (imported from /root/reference) on a seeded synthetic forest and
store inputs + outputs in tests/golden/colorize_golden.npz (the colorize seam may run on the CPU)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_stubs"))
import random
import matplotlib
matplotlib.use("Agg")
from vessel_graph_generation import tree2img as ref

rng = np.random.default_rng(11)
n = 60
a = rng.uniform(0.05, 0.95, (n, 3)); b = a + rng.normal(0, 0.06, (n, 3))
rad = rng.uniform(0.0008, 0.012, n)
forest = [{"node1": a[i], "node2": b[i], "radius": float(rad[i])} for i in range(n)]
out = {"a": a, "b": b, "rad": rad}
for mode in ("continous", "dicrete"):
    random.seed(3)
    img, _ = ref.rasterize_forest(forest, [96, 80], 2, colorize=mode, max_dropout_prob=0.3)
    out[mode] = img.astype(np.uint8)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "colorize_golden.npz"), **out)
print({k: v.shape for k, v in out.items()})
