"""Normalise-on-load convolution (mfma_conv.conv3x3_lazy: whatever kernel the library routes it to) against application pass + plain
convolution on the DynUNet stride-1 shapes, cold cache. Round 5 used it for the in-LDS transform experiment on the DMA-staged kernels
(profiles/r05_unet_lazy_norm1_ab.log); the shipped library routes conv3x3_lazy to the register-staged kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octa_autosegmentation_amd.models import mfma_conv as mc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shapes = [(1216, 32, 32), (608, 64, 64), (304, 128, 128), (152, 256, 256), (152, 512, 512)]
big = torch.empty(400 << 20, dtype=torch.uint8, device="cuda")
def timeit(fn, n=10):
    for _ in range(3): fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        big.zero_(); a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / n
for hw, cin, cout in shapes:
    x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5))
    sc = (1.0 + 0.1 * torch.randn(B, cin, device="cuda")).contiguous(); sh = (0.1 * torch.randn(B, cin, device="cuda")).contiguous()
    with torch.no_grad():
        t_apply = timeit(lambda: mc.materialise((x, sc, sh)))
        xa = mc.materialise((x, sc, sh))
        t_plain = timeit(lambda: mc.conv3x3(xa, w, 1))
        t_lazy = timeit(lambda: mc.conv3x3_lazy((x, sc, sh), w, 1))
        ya, yl = mc.conv3x3(xa, w, 1), mc.conv3x3_lazy((x, sc, sh), w, 1)
        err = (ya.float() - yl.float()).abs().max().item() / ya.float().abs().max().item()
    print(f"{hw:5d}^2 {cin:3d}->{cout:3d}: apply {t_apply*1e3:6.1f} us + conv {t_plain*1e3:6.1f} us = {(t_apply+t_plain)*1e3:6.1f} | lazy conv {t_lazy*1e3:6.1f} us   rel diff {err:.2e}")
