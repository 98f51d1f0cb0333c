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


class DiceBCELoss:
    def __init__(self, sigmoid=False):
        self.bce = torch.nn.BCEWithLogitsLoss() if sigmoid else torch.nn.BCELoss()
        self.dice = DiceLoss(sigmoid=sigmoid)

    def __call__(self, y_pred: torch.Tensor, y: torch.Tensor):
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
