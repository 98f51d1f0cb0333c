"""Model factory of the entry points (reference models/model.py:7-18): `General.model.name` resolved through MODEL_DICT;
ModelInterface classes are instantiated directly, plain networks are wrapped in LambdaModel."""
import torch

from ..utils.enums import Phase
from .lambda_model import LambdaModel
from .model_interface_abc import ModelInterface
from .networks import MODEL_DICT


def define_model(config: dict, phase: Phase):
    device = torch.device(config["General"].get("device") or "cpu")
    model_params: dict = config["General"]["model"]
    model_name = model_params.pop("name")
    model_params["phase"] = phase
    model_params["MODEL_DICT"] = MODEL_DICT
    model_params["inference"] = config["General"].get("inference")
    cls = MODEL_DICT[model_name]
    if isinstance(cls, type) and issubclass(cls, ModelInterface):
        model = cls(**model_params).to(device, non_blocking=True)
    else:
        model = LambdaModel(model_name, **model_params).to(device, non_blocking=True)
    return model
