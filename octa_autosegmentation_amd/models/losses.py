"""Losses on the hot path (reference utils/losses.py:111-121 DiceBCELoss, :183-202 LSGANLoss, :325-353 factory).
MONAI's DiceLoss(sigmoid=True) is restated: per (batch, channel) 1 - (2*sum(p*y) + 1e-5) / (sum(p) + sum(y) + 1e-5), mean."""
import torch


class DiceLoss:
    def __init__(self, sigmoid=False, smooth_nr=1e-5, smooth_dr=1e-5):
        self.sigmoid, self.smooth_nr, self.smooth_dr = sigmoid, smooth_nr, smooth_dr

    def __call__(self, y_pred: torch.Tensor, y: torch.Tensor):
        p = torch.sigmoid(y_pred) if self.sigmoid else y_pred
        dims = tuple(range(2, p.dim()))
        inter = torch.sum(p * y, dim=dims)
        den = torch.sum(p, dim=dims) + torch.sum(y, dim=dims)
        return torch.mean(1.0 - (2.0 * inter + self.smooth_nr) / (den + self.smooth_dr))


class _FusedDiceBCE(torch.autograd.Function):
    """DiceBCELoss(sigmoid=True) on the HIP kernels of csrc/loss.hip: one read of logits + labels forward, one read + one
    write backward (torch: a dozen elementwise / reduction kernels over the 1216 x 1216 map)."""

    @staticmethod
    def forward(ctx, logits, y, smooth_nr, smooth_dr):
        import ctypes
        from .. import _native
        x = logits.contiguous()
        t = y.contiguous().float()
        B = x.shape[0]
        n = x.numel() // B
        sums = torch.empty((B, 4), dtype=torch.float64, device=x.device)
        p = lambda a: ctypes.c_void_p(a.data_ptr())
        rc = _native.lib().octa_dice_bce_fwd(_native.ctx(x.device.index), p(x), 0 if x.dtype == torch.float32 else 1, p(t), B, n, p(sums),
                                             _native.current_stream_ptr())
        _native.check(rc, "octa_dice_bce_fwd")
        # (dice + bce) / 2 with dice = mean(1 - (2 s0 + nr) / (s1 + s2 + dr)), bce = sum(s3) / (B n): one scalar launch, double arithmetic
        loss = torch.empty((), dtype=torch.float32, device=x.device)
        rc = _native.lib().octa_dice_bce_finish(_native.ctx(x.device.index), p(sums), B, n, float(smooth_nr), float(smooth_dr), p(loss),
                                                _native.current_stream_ptr())
        _native.check(rc, "octa_dice_bce_finish")
        ctx.save_for_backward(x, t, sums)
        ctx.smooth = (float(smooth_nr), float(smooth_dr))
        return loss

    @staticmethod
    def backward(ctx, g):
        import ctypes
        from .. import _native
        x, t, sums = ctx.saved_tensors
        B = x.shape[0]
        n = x.numel() // B
        dx = torch.empty_like(x)
        gg = g.reshape(1).float().contiguous()
        p = lambda a: ctypes.c_void_p(a.data_ptr())
        rc = _native.lib().octa_dice_bce_bwd(_native.ctx(x.device.index), p(x), 0 if x.dtype == torch.float32 else 1, p(t), B, n, p(sums), p(gg),
                                             ctx.smooth[0], ctx.smooth[1], p(dx), _native.current_stream_ptr())
        _native.check(rc, "octa_dice_bce_bwd")
        return dx, None, None, None


USE_FUSED_LOSS = True


class DiceBCELoss:
    def __init__(self, sigmoid=False):
        self.sigmoid = sigmoid
        self.bce = torch.nn.BCEWithLogitsLoss() if sigmoid else torch.nn.BCELoss()
        self.dice = DiceLoss(sigmoid=sigmoid)

    def __call__(self, y_pred: torch.Tensor, y: torch.Tensor):
        if (USE_FUSED_LOSS and self.sigmoid and y_pred.is_cuda and y_pred.dtype in (torch.float32, torch.bfloat16)
                and y_pred.shape == y.shape and (y_pred.dim() < 2 or y_pred.shape[1] == 1)):
            return _FusedDiceBCE.apply(y_pred, y, self.dice.smooth_nr, self.dice.smooth_dr)   # one channel: per-sample sums
        return (self.dice(y_pred, y) + self.bce(y_pred, y)) / 2


class LSGANLoss:
    def __init__(self):
        self.loss = torch.nn.MSELoss()

    def __call__(self, prediction: torch.Tensor, target_is_real: bool):
        target = torch.ones_like(prediction) if target_is_real else torch.zeros_like(prediction)
        return self.loss(prediction, target)


def get_loss_function_by_name(name: str, config=None, *args):
    table = {"DiceBCELoss": lambda: DiceBCELoss(True), "LSGANLoss": LSGANLoss,
             "BCELoss": lambda: torch.nn.BCEWithLogitsLoss(), "MSELoss": torch.nn.MSELoss}
    if name not in table:
        raise NotImplementedError(f"loss {name} is outside the MI355X hot path")
    return table[name]()
