"""First-layer (one input channel) kernels at the bench shape: forward (plain / with the statistics epilogue) and weight gradient."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octa_autosegmentation_amd.models import mfma_conv as mc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
x = torch.rand(B, 1216, 1216, 1, device="cuda").to(torch.bfloat16)
w = torch.randn(32, 1, 3, 3, device="cuda", requires_grad=True)
big = torch.empty(400 << 20, dtype=torch.uint8, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        big.zero_()                      # cold Infinity cache, as in the training step
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / n
with torch.no_grad():
    print("c1 fwd ms          ", t(lambda: mc.conv3x3(x, w, 1)))
    print("c1 fwd + stats ms  ", t(lambda: mc.conv3x3(x, w, 1, True)))
y = mc.conv3x3(x, w, 1)
dy = torch.randn_like(y)
print("c1 wgrad ms        ", t(lambda: y.backward(dy, retain_graph=True)))
print("bytes of y: %.0f MB" % (y.numel() * 2 / 1e6))
