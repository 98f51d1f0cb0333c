import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from octa_autosegmentation_amd.models import mfma_conv as mc
x = torch.rand(4, 1216, 1216, 1, device="cuda").to(torch.bfloat16).requires_grad_(False)
w = torch.randn(32, 1, 3, 3, device="cuda", requires_grad=True)
y = mc.conv3x3(x, w, 1)
dy = torch.randn_like(y)
for _ in range(3): y.backward(dy, retain_graph=True)
torch.cuda.synchronize(); t=time.time()
for _ in range(20): y.backward(dy, retain_graph=True)
torch.cuda.synchronize(); print("c1 wgrad ms", (time.time()-t)/20*1e3)
ref = torch.nn.grad.conv2d_weight(x.float().permute(0,3,1,2), (32,1,3,3), dy.float().permute(0,3,1,2), padding=1)
w.grad = None; y.backward(dy, retain_graph=True)
print("max rel err", ((w.grad-ref).abs().max()/ref.abs().max()).item())
