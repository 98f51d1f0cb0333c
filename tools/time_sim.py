"""Quick device timing of the batched simulator (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, yaml
from octa_autosegmentation_amd.vessel_graph_generation import greenhouse
from octa_autosegmentation_amd.utils import configs
cfg = configs.load_generator_config()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
i1 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
i2 = int(sys.argv[3]) if len(sys.argv) > 3 else 150
cfg["Greenhouse"]["modes"][0]["I"] = i1; cfg["Greenhouse"]["modes"][1]["I"] = i2
sim = greenhouse.BatchSimulator(cfg, B)
for rep in range(2):
    torch.cuda.synchronize(); t = time.time()
    res = sim.run(np.arange(B) + 1000 * rep)
    torch.cuda.synchronize(); dt = time.time() - t
    print(f"sim B={B} I={i1}+{i2}: {dt:.3f} s -> {B/dt:.1f} samples/s; edges/sample {np.diff(res.edge_off).mean():.0f}; "
          f"stats err={res.stats[:,0].max()} draws~{res.stats[:,1].mean():.0f} murray~{res.stats[:,2].mean():.0f} bif~{res.stats[:,3].mean():.1f} respec~{res.stats[:,4].mean():.0f}")
    names = ["sample", "assign_art", "pre_art", "seq_art", "satisfy_art", "host_wait", "assign_ven", "pre_ven", "seq_ven", "satisfy_ven"]
    prof = res.stats[:, 8:18].mean(axis=0) / 1e5  # ms
    sub = res.stats[:, 18:24].mean(axis=0) / 1e5
    print("  candidates=%.0f | satisfy_art: kd=%.0f pairs+ven=%.0f sort=%.0f set=%.0f compact=%.0f" % tuple(sub))
    print("  murray walks (both forests): %.0f ms" % (res.stats[:, 31].mean() / 1e5))
    kd = res.stats[:, 24:31].mean(axis=0) / 1e5
    print("  kd: bbox=%.0f dim=%.0f gather=%.0f nth_wave=%.0f nth_thread=%.0f next=%.0f final=%.0f" % tuple(kd))
    print("  timing:", {k: (round(v, 1) if isinstance(v, float) else v) for k, v in res.timing.items()})
    print("  phase ms (mean over samples): " + ", ".join(f"{n}={v:.0f}" for n, v in zip(names, prof) if n != "-") + f"  total={prof.sum():.0f}")
