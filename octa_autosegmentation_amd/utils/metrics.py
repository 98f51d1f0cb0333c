"""Training / validation metrics of the entry points (reference utils/metrics.py:165-196 MetricsManager): DSC and IoU in
the training phase; the validation phase adds AUC, ACC, Recall, Precision and -- when scikit-image is importable -- clDice
(skeletonisation is skimage's; it is not in the MI355X image, so the column is simply absent there). Every score is computed
on the device the tensors live on and read back as ONE scalar per metric and sample; the reference moves whole 1216x1216
maps to numpy for each of them. These are logging quantities, not a kernel target (metrics are out of
scope as kernels)."""
import math

import torch

from .aside import join_aside
from .enums import Phase


class Metric:
    """Scores are kept as 0-dim tensors ON THE DEVICE of the maps (NaN = "skip this sample", e.g. empty ground truth) and meet in
    one stacked nanmean when the epoch ends: a training step queues no host read of its own (the reference reads every score of
    every sample back as a python float)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.scores = []

    def add(self, value):
        self.scores.append(value if torch.is_tensor(value) else torch.tensor(float(value)))

    def aggregate(self) -> torch.Tensor:
        if not self.scores:
            return torch.tensor(float("nan"))
        dev = next((s.device for s in self.scores if s.is_cuda), torch.device("cpu"))
        join_aside(dev)                  # scores of training steps are computed on the side stream (utils/aside.py)
        v = torch.stack([s.detach().to(dev, torch.float64).reshape(()) for s in self.scores])
        ok = ~torch.isnan(v)
        n = ok.sum()
        mean = torch.where(ok, v, torch.zeros_like(v)).sum() / n.clamp(min=1)
        return torch.where(n > 0, mean, torch.full_like(mean, float("nan"))).float().cpu()


def _flat_bool(t):
    return t.detach().reshape(-1) != 0


def _nan_unless(cond, value):
    return torch.where(cond, value, torch.full_like(value, float("nan")))


class MacroDiceMetric(Metric):
    """Per layer 2 |gt & pred| / (|gt| + |pred|) for class 1, NaN (skipped) for an empty ground truth (metrics.py:95-123)."""

    def __call__(self, y_pred, y):
        for p_i, y_i in zip(y_pred, y):
            for layer in range(len(p_i)):
                gt, pr = y_i[layer].detach().float(), p_i[layer].detach().float()
                gsum = gt.sum()
                inter = ((gt == 1) & (pr == 1)).sum()
                self.add(_nan_unless(gsum != 0, 2.0 * inter / (gsum + pr.sum())))

    def aggregate(self):
        return super().aggregate() if self.scores else torch.tensor(0)


class MeanIoU(Metric):
    """MONAI MeanIoU(include_background=True, reduction='mean') on binarised maps: per channel |p & y| / |p | y|; channels with
    an empty ground truth are NaN and ignored (MONAI's ignore_empty default)."""

    def __call__(self, y_pred, y):
        for p_i, y_i in zip(y_pred, y):
            for c in range(len(p_i)):
                p, g = _flat_bool(p_i[c]), _flat_bool(y_i[c])
                self.add(_nan_unless(g.any(), (p & g).sum().double() / (p | g).sum().clamp(min=1).double()))


class _Confusion(Metric):
    def counts(self, p_i, y_i):
        """(tp, tn, fp, fn) as float64 0-dim tensors on the maps' device."""
        p, g = _flat_bool(p_i), _flat_bool(y_i)
        tp, pp, gp = (p & g).sum().double(), p.sum().double(), g.sum().double()
        n = float(p.numel())
        return tp, n - pp - gp + tp, pp - tp, gp - tp


class AccuracyMetric(_Confusion):
    def __call__(self, y_pred, y):
        for p_i, y_i in zip(y_pred, y):
            tp, tn, fp, fn = self.counts(p_i, y_i)
            self.add((tp + tn) / (tp + tn + fp + fn))


class Recall(_Confusion):
    def __call__(self, y_pred, y):
        for p_i, y_i in zip(y_pred, y):
            tp, tn, fp, fn = self.counts(p_i, y_i)
            self.add(_nan_unless(tp + fn > 0, tp / (tp + fn).clamp(min=1)))


class Precision(_Confusion):
    def __call__(self, y_pred, y):
        for p_i, y_i in zip(y_pred, y):
            tp, tn, fp, fn = self.counts(p_i, y_i)
            self.add(_nan_unless(tp + fp > 0, tp / (tp + fp).clamp(min=1)))


class AUCMetric(Metric):
    """ROC AUC of the flattened map (monai.metrics.compute_roc_auc): rank statistic with average ranks for ties."""

    def __call__(self, y_pred, y):
        for p_i, y_i in zip(y_pred, y):
            s, g = p_i.detach().reshape(-1).double(), _flat_bool(y_i)
            n_pos = g.sum().double()
            n_neg = float(g.numel()) - n_pos
            vals, inv, cnt = torch.unique(s, sorted=True, return_inverse=True, return_counts=True)
            hi = torch.cumsum(cnt, 0).double()
            avg_rank = hi - (cnt.double() - 1) / 2          # average 1-based rank of each distinct score
            r_pos = (avg_rank[inv] * g.double()).sum()
            auc = (r_pos - n_pos * (n_pos + 1) / 2) / (n_pos * n_neg).clamp(min=1)
            self.add(_nan_unless((n_pos > 0) & (n_neg > 0), auc))


class ClDiceMetric(Metric):
    def __call__(self, y_pred, y):
        from skimage.morphology import skeletonize
        import numpy as np
        for p_i, y_i in zip(y_pred, y):
            for layer in range(len(p_i)):
                v_p, v_l = p_i[layer].detach().cpu().numpy(), y_i[layer].detach().cpu().numpy()
                cl = lambda v, s: np.sum(v * s) / np.sum(s)
                tprec, tsens = cl(v_p, skeletonize(v_l)), cl(v_l, skeletonize(v_p))
                self.add(float(2 * tprec * tsens / (tprec + tsens)))


def _have_skimage():
    try:
        import skimage.morphology  # noqa: F401
        return True
    except Exception:
        return False


class MetricsManager:
    def __init__(self, phase: Phase = Phase.TRAIN):
        self.metrics = {"DSC": MacroDiceMetric(), "IoU": MeanIoU()}
        if phase != Phase.TRAIN:
            if _have_skimage():
                self.metrics["ClDice"] = ClDiceMetric()
            self.metrics.update({"AUC": AUCMetric(), "ACC": AccuracyMetric(), "Recall": Recall(), "Precision": Precision()})
        self.comp = "DSC"

    def __call__(self, y_pred, y):
        for v in self.metrics.values():
            v(y_pred=y_pred, y=y)

    def aggregate_and_reset(self, prefix: str = ""):
        d = dict()
        for k, v in self.metrics.items():
            d[f"{prefix}_{k}"] = v.aggregate().item()
            v.reset()
        return d

    def get_comp_metric(self, prefix: str):
        return f"{prefix}_{self.comp}"
