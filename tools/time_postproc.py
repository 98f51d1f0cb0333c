"""GPU time of the training loop's per-step extras: post-processing chain and metrics on one 1216^2 prediction."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from octa_autosegmentation_amd.models.postprocess import remove_small_objects_device  # noqa: E402
from octa_autosegmentation_amd.utils.metrics import MetricsManager  # noqa: E402


def timed(f, n=20):
    f(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3


def main():
    torch.cuda.set_device(0)
    g = torch.Generator(device="cuda").manual_seed(0)
    # vessel-like mask: thresholded smooth noise (connected bands) and a noisy early-training prediction
    base = torch.nn.functional.interpolate(torch.rand(1, 1, 76, 76, device="cuda", generator=g), size=(1216, 1216), mode="bilinear")
    smooth = (base > 0.55)
    noisy = smooth ^ (torch.rand(1, 1, 1216, 1216, device="cuda", generator=g) > 0.93)
    for name, m in (("smooth", smooth), ("noisy", noisy), ("pure noise 50%", torch.rand(1, 1, 1216, 1216, device="cuda", generator=g) > 0.5)):
        u8 = m.to(torch.uint8)
        print(f"{name}: foreground {float(m.float().mean()):.2f}  remove_small_objects {timed(lambda: remove_small_objects_device(u8[0], 160, 1)):.3f} ms", flush=True)
    logits = torch.randn(1, 1, 1216, 1216, device="cuda", generator=g).to(torch.bfloat16)
    lab = smooth.float()
    mm = MetricsManager()

    def chain():
        p = torch.sigmoid(logits[0].float())
        p = (p >= 0.5).to(torch.float32)
        keep = remove_small_objects_device((p != 0).to(torch.uint8), 160, 1)
        p = p * keep.to(p.dtype)
        mm(y_pred=[p], y=[lab[0].to(torch.uint8)])
    print(f"post chain + metrics: {timed(chain):.3f} ms per step", flush=True)
    for v in mm.metrics.values():
        v.reset()


if __name__ == "__main__":
    main()
