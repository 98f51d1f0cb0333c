"""Development aid: where a generator thread of the headline loop spends its cycle (bench.py: 4 launches in flight).
  python tools/time_pipeline_stages.py [inflight=4] [steps=6] [batch=128]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from octa_autosegmentation_amd import pipeline, _native
from octa_autosegmentation_amd.utils import configs

n_fly = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
B = int(sys.argv[3]) if len(sys.argv) > 3 else 128
cfg = configs.load_generator_config()
gens = [pipeline.TripleGenerator(cfg, B) for _ in range(n_fly)]
streams = [torch.cuda.Stream() for _ in range(n_fly)]
dev = torch.cuda.current_device()
rows = [[] for _ in range(n_fly)]
spans = []
phase_ms = []


def work(slot):
    torch.cuda.set_device(dev)
    g = gens[slot]
    with torch.cuda.stream(streams[slot]):
        for rep in range(steps):
            seeds = np.arange(B) + 1000 * slot + 100000 * rep
            t0 = time.time()
            res = g.sim.run(seeds)
            t1 = time.time()
            spans.append(res.spans.copy())
            phase_ms.append(res.stats[:, 8:18].sum(axis=1) / 1e5)
            with _native.use_ctx(g._ctx):
                out = g._render(res, True)
            t2 = time.time()
            streams[slot].synchronize()
            t3 = time.time()
            rows[slot].append((t1 - t0, res.timing["loop_wall_ms"] / 1e3, t2 - t1, t3 - t2, t3 - t0))


t = time.time()
ths = [threading.Thread(target=work, args=(i,)) for i in range(n_fly)]
[x.start() for x in ths]; [x.join() for x in ths]
wall = time.time() - t
print(f"{n_fly} in flight, {steps} steps each of {B}: {n_fly * steps * B / wall:.1f} samples/s")
a = np.array([r for rr in rows for r in rr[1:]])
print("per step (s), mean over steps after the first: sim.run %.3f (kernel loop %.3f, rest %.3f) | render enqueue %.3f | render wait %.3f | cycle %.3f"
      % (a[:, 0].mean(), a[:, 1].mean(), (a[:, 0] - a[:, 1]).mean(), a[:, 2].mean(), a[:, 3].mean(), a[:, 4].mean()))
sp = np.concatenate(spans).astype(np.float64) / 1e8
w0, w1 = np.percentile(sp[:, 0], 25), np.percentile(sp[:, 1], 75)      # a window inside the steady state
inside = np.clip(np.minimum(sp[:, 1], w1) - np.maximum(sp[:, 0], w0), 0, None).sum()
cus = torch.cuda.get_device_properties(dev).multi_processor_count
print("CU time used by the simulator in the steady-state window: %.1f %% of %d CUs; span per sample %.1f ms (phase sum %.1f ms)"
      % (100 * inside / (cus * (w1 - w0)), cus, 1e3 * (sp[:, 1] - sp[:, 0]).mean(), np.concatenate(phase_ms).mean()))
