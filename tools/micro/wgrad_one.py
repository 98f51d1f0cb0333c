"""One weight-gradient shape, a few launches (counter passes / traces of a single kernel): python tools/micro/wgrad_one.py HW CIN COUT [B] [N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from octa_autosegmentation_amd.models import mfma_conv
hw, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 4
n = int(sys.argv[5]) if len(sys.argv) > 5 else 5
x = torch.randn(B, hw, hw, cin, device="cuda").to(torch.bfloat16)
dy = torch.randn(B, hw, hw, cout, device="cuda").to(torch.bfloat16)
big = torch.empty(300 << 20, dtype=torch.uint8, device="cuda")
for _ in range(n):
    big.zero_()                       # push x and dy out of the Infinity cache, as the training step's other layers do
    mfma_conv.conv3x3_nhwc_wgrad(x, dy)
torch.cuda.synchronize()
