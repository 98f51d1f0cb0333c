"""Quick device timing of the rasteriser + dither (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from octa_autosegmentation_amd.vessel_graph_generation import tree2img
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "raster_golden.npz"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
es = [g["graph0_edges"], g["graph1_edges"]]
cat = np.concatenate([es[b % 2] for b in range(B)])
off = np.zeros(B + 1, np.int64); off[1:] = np.cumsum([len(es[b % 2]) for b in range(B)])
d = torch.from_numpy(cat).cuda()
for res in ([304, 304], [1216, 1216]):
    out = torch.empty((B, res[1], res[0]), dtype=torch.uint8, device="cuda")
    for it in range(3):
        torch.cuda.synchronize(); t = time.time()
        tree2img.rasterize_edges_device(d, off, res, out=out)
        torch.cuda.synchronize(); dt = time.time() - t
        import ctypes
        from octa_autosegmentation_amd import _native
        prof = np.zeros(4, np.int64); _native.lib().octa_raster_prof(_native.ctx(), prof.ctypes.data)
        print(f"raster {res} B={B}: {dt*1e3:.2f} ms  -> {B/dt:.0f} img/s   WG-ms: bin={prof[1]/1e5:.1f} tess={prof[2]/1e5:.1f} fold={prof[3]/1e5:.1f} err={prof[0]}")
    for it in range(2):
        torch.cuda.synchronize(); t = time.time()
        lab = tree2img.binarize_label_device(out)
        torch.cuda.synchronize(); dt = time.time() - t
        print(f"dither {res} B={B}: {dt*1e3:.2f} ms  -> {B/dt:.0f} img/s")
