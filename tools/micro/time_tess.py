"""GPU: kernel times of one rasterisation (128 graphs at 1216^2 and at 304^2) from torch's profiler-free HIP events around each C-ABI half."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from octa_autosegmentation_amd.vessel_graph_generation import tree2img
g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "raster_golden.npz"))
B = 128
es = [g["graph0_edges"], g["graph1_edges"]]
cat = np.concatenate([es[b % 2] for b in range(B)])
off = np.zeros(B + 1, np.int64); off[1:] = np.cumsum([len(es[b % 2]) for b in range(B)])
d = torch.from_numpy(cat).cuda()
for res in ([304, 304], [1216, 1216]):
    best = [1e9, 1e9]
    for it in range(5):
        torch.cuda.synchronize(); t0 = time.time()
        plan = tree2img.rasterize_edges_device_plan(d, off, res)
        torch.cuda.synchronize(); t1 = time.time()
        out = tree2img.rasterize_edges_device_draw(plan)
        torch.cuda.synchronize(); t2 = time.time()
        best = [min(best[0], t1 - t0), min(best[1], t2 - t1)]
    print(f"{res}: plan {best[0]*1e3:.2f} ms, draw {best[1]*1e3:.2f} ms")
