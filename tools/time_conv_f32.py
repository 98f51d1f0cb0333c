"""Time the single-precision convolutions of DynUNet-S at 1 x 1 x 1216 x 1216 (the test.py / validate.py pass), layer by layer:
csrc/conv_f32.hip against the vendor library's fp32 convolution on the same inputs. Usage: python tools/time_conv_f32.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octa_autosegmentation_amd.models import conv_f32, networks  # noqa: E402


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    torch.manual_seed(0)
    torch.backends.cudnn.allow_tf32 = False
    torch.backends.cuda.matmul.allow_tf32 = False
    net = networks.DynUNet().cuda().eval()
    shapes = {}
    hooks = []
    for name, m in net.named_modules():
        if isinstance(m, (torch.nn.Conv2d, torch.nn.ConvTranspose2d)):
            hooks.append(m.register_forward_pre_hook(lambda mod, inp, name=name: shapes.__setitem__(name, tuple(inp[0].shape))))
    x = torch.rand(1, 1, 1216, 1216, device="cuda")
    old = networks.USE_F32_MFMA
    networks.USE_F32_MFMA = False
    with torch.no_grad():
        net(x)
    networks.USE_F32_MFMA = old
    for h in hooks:
        h.remove()
    tot_o = tot_v = tot_f = 0.0
    mods = dict(net.named_modules())
    print(f"{'layer':44s} {'input':>22s} {'k/s':>4s} {'GFLOP':>7s} {'ours ms':>8s} {'TF/s':>6s} {'vendor ms':>9s}")
    with torch.no_grad():
        for name, shp in shapes.items():
            m = mods[name]
            xi = torch.randn(*shp, device="cuda")
            yo = conv_f32.forward(m, xi)
            yv = m(xi) if not networks.USE_F32_MFMA else torch.nn.functional.conv2d(xi, m.weight, m.bias, m.stride, m.padding) \
                if isinstance(m, torch.nn.Conv2d) else torch.nn.functional.conv_transpose2d(xi, m.weight, None, m.stride)
            err = (yo - yv).abs().max().item() / max(yv.abs().max().item(), 1e-9)
            to = timed(lambda: conv_f32.forward(m, xi))
            if isinstance(m, torch.nn.Conv2d):
                tv = timed(lambda: torch.nn.functional.conv2d(xi, m.weight, m.bias, m.stride, m.padding))
            else:
                tv = timed(lambda: torch.nn.functional.conv_transpose2d(xi, m.weight, None, m.stride))
            fl = 2.0 * yo.numel() * m.in_channels * (m.kernel_size[0] ** 2 if isinstance(m, torch.nn.Conv2d) else 1) / 1e9
            tot_o += to; tot_v += tv; tot_f += fl
            print(f"{name:44s} {str(shp):>22s} {m.kernel_size[0]}/{m.stride[0]:<2d} {fl:7.2f} {to:8.3f} {fl / to:6.1f} {tv:9.3f}   rel.diff {err:.1e}")
    print(f"total: {tot_f:.1f} GFLOP   ours {tot_o:.2f} ms ({tot_f / tot_o:.1f} TFLOP/s)   vendor {tot_v:.2f} ms")
    with torch.no_grad():
        t = timed(lambda: net(x), reps=5)
    print(f"whole fp32 inference pass (convolutions + InstanceNorm/LeakyReLU + concatenations): {t:.2f} ms")


if __name__ == "__main__":
    main()
