"""Kernel-level timing of the MFMA 3x3 convolution on the DynUNet-S layer shapes (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST"); os.environ.setdefault("MIOPEN_DEBUG_CONV_GEMM", "0")
import torch, torch.nn.functional as F
from octa_autosegmentation_amd.models import mfma_conv
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shapes = [(1216, 32, 32, 1), (1216, 64, 32, 1), (1216, 32, 64, 2), (608, 64, 64, 1), (608, 128, 64, 1), (608, 64, 128, 2),
          (304, 128, 128, 1), (304, 256, 128, 1), (304, 128, 256, 2), (152, 256, 256, 1), (152, 512, 256, 1), (152, 256, 512, 1), (152, 512, 512, 1)]
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t) / n
tot_m = tot_t = 0
for hw, cin, cout, st in shapes:
    x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") / (3 * cin ** 0.5)).to(torch.bfloat16)
    wt = mfma_conv.pack_weight(w)
    xn = x.permute(0, 3, 1, 2).contiguous()
    t_m = timeit(lambda: mfma_conv.conv3x3_nhwc(x, wt, stride=st))
    ho_ = (hw - 1) // st + 1
    dy = torch.randn(B, ho_, ho_, cout, device="cuda").to(torch.bfloat16)
    wd = mfma_conv.pack_weight_dgrad(w)
    t_d = timeit(lambda: mfma_conv.conv3x3_nhwc(dy, wd, stride=1, in_dilation=st))
    t_w = timeit(lambda: mfma_conv.conv3x3_nhwc_wgrad(x, dy)) if st == 1 and not os.environ.get("OCTA_SKIP_WGRAD") else float("nan")
    t_t = timeit(lambda: F.conv2d(xn, w, stride=st, padding=1)) if not os.environ.get("OCTA_SKIP_TORCH") else float("nan")
    ho = (hw - 1) // st + 1
    fl = 2.0 * 9 * cin * cout * ho * ho * B
    by = 2.0 * B * (hw * hw * cin + ho * ho * cout)
    tot_m += t_m; tot_t += t_t
    print(f"   dgrad {t_d*1e3:7.3f} ms {2.0*9*cin*cout*ho_*ho_*B/t_d/1e12:6.1f} TF/s | wgrad {t_w*1e3:7.3f} ms {2.0*9*cin*cout*ho_*ho_*B/t_w/1e12:6.1f} TF/s")
    print(f"{hw:5d}^2 {cin:3d}->{cout:3d} s{st}: mfma {t_m*1e3:7.3f} ms {fl/t_m/1e12:6.1f} TF/s {by/t_m/1e9:6.0f} GB/s | torch/MIOpen NCHW {t_t*1e3:7.3f} ms {fl/t_t/1e12:6.1f} TF/s")
print(f"sum: mfma {tot_m*1e3:.2f} ms, torch {tot_t*1e3:.2f} ms")
