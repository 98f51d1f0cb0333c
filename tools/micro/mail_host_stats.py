import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from octa_autosegmentation_amd.utils import configs
from octa_autosegmentation_amd.vessel_graph_generation import greenhouse
sim = greenhouse.BatchSimulator(configs.load_generator_config(), 512)
for rep in range(2):
    res = sim.run(np.arange(512) + 5000 + 1000 * rep)
print({k: (round(v, 2) if isinstance(v, float) else v) for k, v in res.timing.items()})
sim.close()
