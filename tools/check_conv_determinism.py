"""Run-to-run bit equality of the MFMA convolution at the U-Net's layer sizes, under load (development aid: a race in the LDS
staging would show up as a differing output)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octa_autosegmentation_amd.models import mfma_conv as mc
torch.manual_seed(0)
bad = 0
for (n, h, w, cin, cout, dil, mask) in [(4, 152, 152, 512, 512, 1, 0x1ff), (4, 304, 304, 128, 128, 1, 0x1ff), (4, 608, 608, 64, 64, 1, 0x1ff),
                                        (4, 1216, 1216, 32, 32, 1, 0x1ff), (4, 1216, 1216, 64, 32, 1, 0x1ff), (4, 304, 304, 128, 64, 2, 0x1ff),
                                        (4, 304, 304, 128, 64, 2, 0x1b0), (3, 77, 45, 96, 160, 1, 0x1ff)]:
    x = torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16)
    wgt = (torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5))
    wt = mc.pack_weight(wgt)
    y0 = mc.conv3x3_nhwc(x, wt, 1, dil, mask)
    diffs = 0
    for i in range(30):
        y = mc.conv3x3_nhwc(x, wt, 1, dil, mask)
        if not torch.equal(y, y0):
            diffs += 1
            d = (y.float() - y0.float()).abs()
            print("   run", i, "differs at", int((d > 0).sum()), "elements, max", d.max().item())
    bad += diffs
    print(f"N{n} {h}x{w} {cin}->{cout} dil{dil} mask{mask:#x}: {diffs} of 30 runs differ")
print("RESULT", "DETERMINISTIC" if bad == 0 else "NONDETERMINISTIC")
