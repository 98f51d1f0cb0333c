"""ModelInterface wrapper of a plain network from MODEL_DICT -- how the segmentation configs train DynUNet
(reference models/lambda_model.py:13-71): one optimiser over `model`, loss `Train.loss`, outputs/labels of sample 0
post-processed for the metrics."""
from typing import Any, Callable, Dict, Tuple

import torch

from ..utils.enums import Phase
from .base_model_abc import BaseModelABC, aside
from .losses import get_loss_function_by_name
from .model_interface_abc import Output


def decollate_batch(t: torch.Tensor):
    """What monai.data.decollate_batch does to a batched tensor: a list of its leading-dimension slices."""
    return [t[i] for i in range(t.shape[0])]


class LambdaModel(BaseModelABC):
    def __init__(self, model_name: str, phase: Phase, MODEL_DICT: dict, inference: str = "model", **kwargs) -> None:
        super().__init__(optimizer_mapping={"optimizer": ["model"]})
        self.model = MODEL_DICT[model_name](**kwargs)

    def initialize_model_and_optimizer(self, init_mini_batch: dict, init_weights: Callable, config: Dict[str, dict], args, scaler,
                                       phase: Phase = Phase.TRAIN) -> None:
        if phase != Phase.TEST:
            self.loss_name = config.get(Phase.TRAIN, dict()).get("loss", "")
            self.loss_function = get_loss_function_by_name(self.loss_name, config)
        if phase == Phase.TRAIN and config[Phase.TRAIN].get("AT", False):
            raise NotImplementedError("adversarial training (Train.AT, configs *_RA / *_AA) is outside the MI355X hot path")
        super().initialize_model_and_optimizer(init_mini_batch, init_weights, config, args, scaler, phase)

    def inference(self, mini_batch: Dict[str, Any], post_transformations: Dict[str, Callable], device: torch.device = "cpu",
                  phase: Phase = Phase.TEST) -> Tuple[Output, Dict[str, torch.Tensor]]:
        inputs = mini_batch["image"].to(device, non_blocking=True)
        labels = mini_batch["label"].to(device, non_blocking=True) if phase != Phase.TEST else None
        pred = self.model(inputs).squeeze(-1)
        if phase == Phase.TRAIN:          # the scored sample's post-processing leaves the training stream (base_model_abc.aside)
            with aside(pred.device, pred, labels):
                outputs: Output = {"prediction": [post_transformations["prediction"](i) for i in decollate_batch(pred.detach()[0:1])],
                                   "label": [post_transformations["label"](i) for i in decollate_batch(labels[0:1])]}
            return outputs, {self.loss_name: self.loss_function(y_pred=pred.float(), y=labels.float())}
        outputs: Output = {"prediction": [post_transformations["prediction"](i) for i in decollate_batch(pred[0:1])]}
        if phase != Phase.TEST:
            outputs["label"] = [post_transformations["label"](i) for i in decollate_batch(labels[0:1])]
            losses = {self.loss_name: self.loss_function(y_pred=pred.float(), y=labels.float())}
        else:
            losses = None
        return outputs, losses

    def forward(self, input: torch.Tensor) -> torch.Tensor:
        return self.model(input)

    def compute_metric(self, outputs: Output, metrics) -> None:
        t = outputs["prediction"][0]
        if self.training and torch.is_tensor(t):
            with aside(t.device):
                metrics(outputs["prediction"], outputs["label"])
        else:
            metrics(outputs["prediction"], outputs["label"])

    def plot_sample(self, visualizer, mini_batch: Dict[str, Any], outputs: Output, *, suffix: str = "") -> str:
        return visualizer.plot_sample(mini_batch["image"][0], outputs["prediction"][0], outputs["label"][0], suffix=suffix)
