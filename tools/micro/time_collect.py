"""GPU: where the time of BatchSimulator._collect goes (it runs inside the generator pipeline's gate, between two persistent kernels)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from octa_autosegmentation_amd import _native
from octa_autosegmentation_amd.utils import configs
from octa_autosegmentation_amd.vessel_graph_generation import greenhouse as gh

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sim = gh.BatchSimulator(configs.load_generator_config(), B)
lib = sim._lib
for rep in range(2):
    seeds = np.ascontiguousarray(np.arange(B) + 100 + 1000 * rep, dtype=np.uint32); py = seeds.astype(np.uint64)
    t = [time.time()]
    rc = lib.octa_sim_run(sim._h, seeds.ctypes.data, py.ctypes.data, sim._bif_fn, None, _native.current_stream_ptr()); t.append(time.time())
    off = np.zeros(B + 1, np.int64); n_art = np.zeros(B, np.int64)
    lib.octa_sim_edge_offsets(sim._h, off.ctypes.data, n_art.ctypes.data); t.append(time.time())
    d_edges = torch.empty((int(off[-1]), 7), dtype=torch.float64, device="cuda"); t.append(time.time())
    lib.octa_sim_export_edges_device(sim._h, ctypes.c_void_p(d_edges.data_ptr()), _native.current_stream_ptr()); t.append(time.time())
    stats = np.zeros((B, 32), np.int64); lib.octa_sim_stats(sim._h, stats.ctypes.data); t.append(time.time())
    timing = np.zeros(8); lib.octa_sim_timing(sim._h, timing.ctypes.data); svc = np.zeros(5); lib.octa_sim_service_stats(sim._h, svc.ctypes.data); t.append(time.time())
    spans = np.zeros((B, 2), np.int64); lib.octa_sim_spans(sim._h, spans.ctypes.data); t.append(time.time())
    torch.cuda.synchronize(); t.append(time.time())
    names = ["octa_sim_run", "edge_offsets", "torch.empty", "export launch", "stats", "timing+service", "spans", "sync (export kernel)"]
    print(f"rep {rep}: kernel {timing[2]:.1f} ms; " + ", ".join(f"{n} {1e3 * (b - a):.2f}" for n, a, b in zip(names, t[:-1], t[1:])) + " ms")
