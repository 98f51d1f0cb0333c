"""Development aid: split-K batched GEMM for the 1x1 weight gradient (fp32 output) vs the tap-masked MFMA kernel."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from octa_autosegmentation_amd.models import mfma_conv as mc
n, h, w, cin, cout = 4, 152, 152, 512, 256
x = torch.randn(n, h, w, cin, device="cuda").to(torch.bfloat16)
dy = torch.randn(n, h, w, cout, device="cuda").to(torch.bfloat16)
M = n * h * w
ref = (x.reshape(M, cin).float().t() @ dy.reshape(M, cout).float())
def t(f, name):
    for _ in range(3): r = f()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(20): r = f()
    torch.cuda.synchronize(); dt = (time.time() - t0) / 20
    err = (r.float() - ref).abs().max().item() / ref.abs().max().item()
    print(f"{name}: {dt*1e3:.3f} ms, rel err {err:.2e}", flush=True)
t(lambda: mc.conv3x3_nhwc_wgrad(x, dy, tap_mask=1 << 4)[:, :, 1, 1].t(), "masked mfma wgrad")
for S in (8, 16, 32, 64):
    if M % S: continue
    x2, d2 = x.reshape(S, M // S, cin), dy.reshape(S, M // S, cout)
    try:
        t(lambda: torch.bmm(x2.transpose(1, 2), d2, out_dtype=torch.float32).sum(0), f"bmm out_dtype fp32 S={S}")
    except Exception as e:
        print("bmm out_dtype failed:", repr(e)[:200])
        break
for S in (16,):
    x2, d2 = x.reshape(S, M // S, cin), dy.reshape(S, M // S, cout)
    t(lambda: torch.bmm(x2.transpose(1, 2), d2).float().sum(0), f"bmm bf16 out S={S}")
