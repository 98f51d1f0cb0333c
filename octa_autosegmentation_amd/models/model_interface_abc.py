"""The contract train.py / test.py / validate.py program against (reference models/model_interface_abc.py:18-137): same
method names, argument meaning and return conventions, so the reference's entry points drive the MI355X models unchanged.

Output is a dict with "prediction" (list of post-processed tensors, one per decollated sample) and optionally "label";
losses are a dict name -> value (python floats from perform_training_step, tensors from inference)."""
from abc import ABC, abstractmethod
from typing import Any, Callable, Dict, List, Tuple, TypedDict

import torch

from ..utils.enums import Phase


class Output(TypedDict, total=False):
    prediction: List[torch.Tensor]
    label: List[torch.Tensor]


class ModelInterface(ABC):
    @abstractmethod
    def initialize_model_and_optimizer(self, init_mini_batch: dict, init_weights: Callable, config: dict, args, scaler,
                                       phase: Phase = Phase.TRAIN) -> None:
        """Create optimisers / schedulers and initialise or load weights (checkpoint naming: base_model_abc.py)."""

    @abstractmethod
    def eval(self):
        ...

    @abstractmethod
    def train(self):
        ...

    @abstractmethod
    def compute_metric(self, outputs: Output, metrics) -> None:
        """Feed one mini-batch's outputs to the MetricsManager."""

    @abstractmethod
    def forward(self, input: torch.Tensor) -> torch.Tensor:
        """Minimal forward pass of the underlying network."""

    @abstractmethod
    def inference(self, mini_batch: Dict[str, Any], post_transformations: Dict[str, Callable], device: torch.device = "cpu",
                  phase: Phase = Phase.TEST) -> Tuple[Output, Dict[str, torch.Tensor]]:
        """Full forward pass of a mini-batch: (outputs, losses as tensors; None in the test phase)."""

    @abstractmethod
    def perform_training_step(self, mini_batch: Dict[str, Any], scaler, post_transformations: Dict[str, Callable],
                              device: torch.device = "cpu") -> Tuple[Output, Dict[str, float]]:
        """One optimiser step on a mini-batch: (outputs, losses as floats)."""

    @abstractmethod
    def plot_sample(self, visualizer, mini_batch: Dict[str, Any], outputs: Output, *, suffix: str = "") -> str:
        """Write a sample figure through the visualizer; returns its path."""
