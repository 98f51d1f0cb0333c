"""Development aid: per-kernel time of the one-channel-side convolutions at the GAN networks' shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octa_autosegmentation_amd.models import thin_conv as tc


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for n in (2, 4, 8):
    for name, k, pad, c, h in (("G stem", 7, 0, 64, 310), ("D stem", 4, 1, 64, 304)):
        w = torch.randn(c, k * k, device="cuda"); bias = torch.randn(c, device="cuda")
        x = torch.randn(n, h, h, device="cuda").bfloat16()
        y = tc._expand(x, w, bias, k, pad, False, 1.0)
        dy = torch.randn_like(y)
        print(f"N={n} {name}: fwd(expand) {timed(lambda: tc._expand(x, w, bias, k, pad, False, 1.0)):7.1f} us | dx(squeeze) "
              f"{timed(lambda: tc._squeeze(dy, w, None, k, k - 1 - pad, True)):7.1f} us | dw(wgrad) {timed(lambda: tc._wgrad(dy, x, k, pad, False, True)):7.1f} us")
    for name, k, pad, c, h in (("G head", 7, 0, 64, 310), ("D head", 4, 1, 512, 37)):
        w = torch.randn(c, k * k, device="cuda"); bias = torch.randn(1, device="cuda")
        x = torch.randn(n, h, h, c, device="cuda").bfloat16()
        y = tc._squeeze(x, w, bias, k, pad, False)
        dy = torch.randn_like(y)
        print(f"N={n} {name}: fwd(squeeze) {timed(lambda: tc._squeeze(x, w, bias, k, pad, False)):7.1f} us | dx(expand) "
              f"{timed(lambda: tc._expand(dy, w, None, k, k - 1 - pad, True, 1.0)):7.1f} us | dw(wgrad) {timed(lambda: tc._wgrad(x, dy, k, k - 1 - pad, True, False)):7.1f} us")
