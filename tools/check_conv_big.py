"""MFMA convolution against torch's fp32 convolution at the U-Net's real layer sizes (development aid; the unit tests use small shapes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octa_autosegmentation_amd.models import mfma_conv as mc
torch.manual_seed(0)
for (n, h, w, cin, cout, dil) in [(2, 152, 152, 512, 512, 1), (2, 152, 152, 256, 512, 1), (1, 304, 304, 128, 128, 1), (1, 608, 608, 64, 64, 1),
                                  (1, 1216, 1216, 32, 32, 1), (1, 1216, 1216, 64, 32, 1), (1, 304, 304, 128, 64, 2), (3, 77, 45, 96, 160, 1)]:
    x = torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16)
    wgt = (torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5))
    y = mc.conv3x3_nhwc(x, mc.pack_weight(wgt), 1, dil).float()
    xr = x.float().permute(0, 3, 1, 2)
    if dil == 2:
        z = torch.zeros(n, cin, 2 * h, 2 * w, device="cuda"); z[:, :, ::2, ::2] = xr; xr = z
    ref = torch.nn.functional.conv2d(xr, wgt.to(torch.bfloat16).float(), padding=1).permute(0, 2, 3, 1)
    err = (y - ref).abs().max().item()
    print(f"N{n} {h}x{w} {cin}->{cout} dil{dil}: max err {err:.4f} (ref max {ref.abs().max().item():.2f}) {'OK' if err < 0.03 * max(ref.abs().max().item(), 1) else 'MISMATCH'}")
print("weight gradient:")
for (n, h, w, cin, cout) in [(2, 152, 152, 256, 256), (1, 304, 304, 128, 128), (1, 608, 608, 64, 64), (1, 1216, 1216, 32, 32), (1, 1216, 1216, 64, 32), (2, 100, 75, 96, 64)]:
    x = torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16)
    dy = torch.randn(n, h, w, cout, device="cuda").to(torch.bfloat16)
    for rep in range(3):
        dw = mc.conv3x3_nhwc_wgrad(x, dy)
        ref = torch.nn.grad.conv2d_weight(x.float().permute(0, 3, 1, 2), (cout, cin, 3, 3), dy.float().permute(0, 3, 1, 2), padding=1)
        err = (dw - ref).abs().max().item()
        ok = err < 2e-3 * ref.abs().max().item()
        if rep == 0 or not ok:
            print(f"N{n} {h}x{w} {cin}->{cout}: max err {err:.4f} (ref max {ref.abs().max().item():.1f}) {'OK' if ok else 'MISMATCH'}")
