"""GPU only: repeat a 512-sample batch; for every sample whose per-iteration trace differs from the first run's, print the first
iteration that differs and the counts around it."""
import sys, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octa_autosegmentation_amd.utils import configs
from octa_autosegmentation_amd.vessel_graph_generation import greenhouse
cfg = configs.load_generator_config()
sim = greenhouse.BatchSimulator(cfg, 512)
seeds = np.arange(512) + 90000
ref = None
nrep = int(sys.argv[1]) if len(sys.argv) > 1 else 30
events = 0
for rep in range(nrep):
    res = sim.run(seeds)
    tr = sim.trace().copy()
    if ref is None:
        ref = tr
        continue
    bad = np.flatnonzero((tr != ref).any(axis=(1, 2)))
    for k in bad:
        it = int(np.flatnonzero((tr[k] != ref[k]).any(axis=1))[0])
        events += 1
        print(f"rep {rep} sample {k}: first differing iteration {it}: ref {ref[k][max(it-1,0):it+2].tolist()} got {tr[k][max(it-1,0):it+2].tolist()}", flush=True)
print("events", events, "in", (nrep - 1) * 512, "sample runs")
sim.close()
