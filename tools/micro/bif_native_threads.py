import ctypes, os, sys, time, threading
import numpy as np
sys.path.insert(0, "/root/repo")
from octa_autosegmentation_amd import _native
from octa_autosegmentation_amd.utils import configs
from octa_autosegmentation_amd.vessel_graph_generation import greenhouse as gh
cfg = configs.load_generator_config()
fn = gh._enable_native_bifurcation_service(cfg)
g = cfg["Greenhouse"]; r = g["r"] / g["param_scale"]
kappas = tuple(sorted({float(m["kappa"]) for m in g["modes"]}))
rng = np.random.default_rng(7)
m = 20000
recs = np.zeros((m, gh._REC_DOUBLES)); counts = np.zeros(m, np.int32)
for i in range(m):
    n = int(rng.integers(2, 12)); counts[i] = n
    pos = rng.uniform(0.1, 0.9, 3) * np.array([1, 1, 0.0131])
    atts = pos + rng.normal(0, rng.uniform(0.002, 0.08), (n, 3)) * np.array([1, 1, 0.05])
    recs[i, 1:4] = pos; recs[i, 4:7] = [r, kappas[i % len(kappas)], rng.uniform(0.012, 0.034)]; recs[i, 7:7 + 3 * n] = atts.ravel()
recs.view(np.int32).reshape(m, -1)[:, 1] = counts
lib = _native.lib()
for T in (1, 2, 4):
    gots = [np.zeros((m, 6)) for _ in range(T)]
    def work(k):
        for rep in range(3):
            lib.octa_bif_native(m, recs.ctypes.data, gots[k].ctypes.data, None)
    th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
    t = time.time(); [x.start() for x in th]; [x.join() for x in th]; dt = time.time() - t
    print(f"{T} threads: {dt / (3 * m) * 1e6:.2f} us per request per thread; equal {all((g_ == gots[0]).all() for g_ in gots)}")
