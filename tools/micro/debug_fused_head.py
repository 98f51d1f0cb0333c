"""Development aid: per-tensor errors of the fused norm+head against torch fp32."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, torch.nn.functional as F
from octa_autosegmentation_amd.models import mfma_conv
g = torch.Generator(device="cuda").manual_seed(69)
n, h, w, c = 2, 37, 53, 32
x = (torch.randn(n, h, w, c, device="cuda", generator=g) * 1.7 + 0.3).to(torch.bfloat16)
gamma = torch.rand(c, device="cuda", generator=g) + 0.5
beta = torch.randn(c, device="cuda", generator=g) * 0.2
hw_ = torch.randn(1, c, 1, 1, device="cuda", generator=g) / c ** 0.5
hb = torch.randn(1, device="cuda", generator=g)
dl = torch.randn(n, h, w, 1, device="cuda", generator=g).to(torch.bfloat16)
xr, gr, br, wr, hbr = (t.clone().double().requires_grad_(True) for t in (x, gamma, beta, hw_, hb))
y = F.leaky_relu(F.instance_norm(xr.permute(0, 3, 1, 2), weight=gr, bias=br, eps=1e-5), 0.01)
lr = (y.permute(0, 2, 3, 1) @ wr.reshape(-1, 1) + hbr)
lr.backward(dl.double())
xm, gm, bm, wm, hbm = (t.clone().requires_grad_(True) for t in (x, gamma, beta, hw_, hb))
lm = mfma_conv.instance_norm_leaky_relu_head1_nhwc(xm, gm, bm, 0.01, 1e-5, wm, hbm)
lm.backward(dl)
# unfused route for comparison
xu, gu, bu, wu, hbu = (t.clone().requires_grad_(True) for t in (x, gamma, beta, hw_, hb))
lu = mfma_conv.conv1x1_bias_nhwc(mfma_conv.instance_norm_leaky_relu_nhwc(xu, gu, bu, 0.01, 1e-5), wu, hbu)
lu.backward(dl)
for name, a, u, b in (("logits", lm, lu, lr), ("dx", xm.grad, xu.grad, xr.grad), ("dgamma", gm.grad, gu.grad, gr.grad), ("dbeta", bm.grad, bu.grad, br.grad),
                      ("dheadw", wm.grad, wu.grad, wr.grad), ("dheadb", hbm.grad, hbu.grad, hbr.grad)):
    s = b.abs().max().item()
    print(f"{name}: fused err {(a.float()-b.float()).abs().max().item()/s:.2e}  unfused err {(u.float()-b.float()).abs().max().item()/s:.2e}  scale {s:.3g}")
