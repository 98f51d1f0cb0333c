import sys, numpy as np
sys.path.insert(0, "/root/repo")
from octa_autosegmentation_amd.utils import configs
from octa_autosegmentation_amd.vessel_graph_generation import greenhouse
cfg = configs.load_generator_config()
for m in cfg["Greenhouse"]["modes"]: m["I"] = 3
a = greenhouse.BatchSimulator(cfg, 512); b = greenhouse.BatchSimulator(cfg, 512)
for name, sim in (("A", a), ("B", b), ("A", a), ("B", b)):
    print("run", name, file=sys.stderr); sim.run(np.arange(512))
