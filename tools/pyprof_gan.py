"""Development aid: host-side (Python) profile of the GAN-seg training step -- where the launch gaps come from."""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "4"]
import torch
import runpy
ns = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "time_gan.py"))
tr, batch = ns["tr"], ns["batch"]
pr = cProfile.Profile()
torch.cuda.synchronize()
pr.enable()
for _ in range(10):
    tr.perform_training_step(batch)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
