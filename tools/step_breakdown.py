import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, yaml
from octa_autosegmentation_amd import graph_io, pipeline
from octa_autosegmentation_amd.vessel_graph_generation import tree2img
g = np.load("/root/repo/tests/golden/sim_golden.npz"); cfg = yaml.safe_load(str(g["config_yaml"]))
B = 128
gen = pipeline.TripleGenerator(cfg, B)
for rep in range(2):
    t0 = time.time(); res = gen.sim.run(np.arange(B) + 50 * rep); t1 = time.time()
    d_edges = torch.from_numpy(res.edges).cuda(); torch.cuda.synchronize(); t2 = time.time()
    rb = graph_io.edges_as_read_back(res.edges); t3 = time.time()
    d_rb = torch.from_numpy(rb).cuda(); torch.cuda.synchronize(); t4 = time.time()
    off, n_art = res.edge_off, res.n_art
    split = np.empty(2 * B + 1, np.int64); split[0::2] = off; split[1::2] = off[:-1] + n_art
    pair = tree2img.rasterize_edges_device(d_edges, split, [304, 304], 2); grey = tree2img.rasterize_edges_device(d_rb, off, [1216, 1216], 2)
    lab = tree2img.binarize_label_device(grey); torch.cuda.synchronize(); t5 = time.time()
    print("sim.run %.0f ms (loop %.0f, bif %.0f) | H2D edges %.0f | read_back %.0f | H2D rb %.0f | raster+dither %.0f" % (
        (t1 - t0) * 1e3, res.timing["loop_wall_ms"], res.timing["host_bif_ms"], (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3))
