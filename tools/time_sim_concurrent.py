"""Per-phase device time of the simulator with K launches of B samples in flight (development aid): shows which phases get
slower when more workgroups than one launch's share the GPU.   python tools/time_sim_concurrent.py [K=4] [B=128]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from octa_autosegmentation_amd.utils import configs
from octa_autosegmentation_amd.vessel_graph_generation import greenhouse
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cfg = configs.load_generator_config()
names = ["sample", "assign_art", "pre_art", "seq_art", "satisfy_art", "host_wait", "assign_ven", "pre_ven", "seq_ven", "satisfy_ven"]
for k in sorted({1, K}):
    sims = [greenhouse.BatchSimulator(cfg, B) for _ in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    out = [None] * k
    def work(i, reps):
        torch.cuda.set_device(0)
        with torch.cuda.stream(streams[i]):
            for r in range(reps):
                out[i] = sims[i].run(np.arange(B) + 1000 * i + 100000 * r)
    for reps in (1, 3):
        t0 = time.time()
        ths = [threading.Thread(target=work, args=(i, reps)) for i in range(k)]
        [t.start() for t in ths]; [t.join() for t in ths]
        dt = time.time() - t0
    prof = np.mean([o.stats[:, 8:18].mean(axis=0) for o in out], axis=0) / 1e5
    sub = np.mean([o.stats[:, 18:24].mean(axis=0) for o in out], axis=0) / 1e5
    print(f"{k} launch(es) of {B} in flight: {k * B * 3 / dt:.0f} samples/s, kernel {np.mean([o.timing['kernel_b_ms'] for o in out]):.0f} ms; per-sample phase ms: "
          + ", ".join(f"{n}={v:.0f}" for n, v in zip(names, prof)) + f" total={prof.sum():.0f}; candidates={sub[0]:.0f} kd={sub[1]:.0f} set={sub[4]:.0f}")
    for s in sims:
        s.close()
