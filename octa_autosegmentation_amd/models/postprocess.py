"""Inference post-processing on the GPU: the `post_processing.prediction` list of the configs
(configs/config_ves_seg-S.yml:103-113: Activations(sigmoid) -> AsDiscrete(0.5) -> RemoveSmallObjects(160), label:
CastToType uint8) as used by test.py / validate.py, for a whole batch of logits in HBM."""
import ctypes

import torch

from .. import _native


def remove_small_objects_device(mask, min_size=64, connectivity=1, on_value=1):
    """mask: CUDA uint8 / bool [B,H,W] or [B,1,H,W] (non-zero = foreground) -> uint8 of the same shape with every
    connected component smaller than min_size removed (skimage.morphology.remove_small_objects semantics)."""
    assert mask.is_cuda
    shape = mask.shape
    m = mask.reshape(-1, shape[-2], shape[-1]).to(torch.uint8).contiguous()
    out = torch.empty_like(m)
    rc = _native.lib().octa_remove_small_objects(_native.ctx(m.device.index), ctypes.c_void_p(m.data_ptr()), m.shape[0], m.shape[1], m.shape[2],
                                                 int(min_size), int(connectivity), int(on_value), ctypes.c_void_p(out.data_ptr()),
                                                 _native.current_stream_ptr())
    _native.check(rc, "octa_remove_small_objects")
    return out.view(shape)


def postprocess_prediction(logits, post_config):
    """Apply the reference's `post_processing.prediction` list (names as in the YAML) to CUDA logits [B,1,H,W]."""
    x = logits
    for d in post_config:
        name = d["name"]
        if name == "Activations":
            if d.get("sigmoid", False):
                x = torch.sigmoid(x.float())
            elif d.get("softmax", False):
                x = torch.softmax(x.float(), dim=1)
        elif name == "AsDiscrete":
            if "threshold" in d:
                x = (x >= float(d["threshold"])).to(torch.uint8)
            elif d.get("argmax", False):
                x = x.argmax(dim=1, keepdim=True).to(torch.uint8)
        elif name == "RemoveSmallObjects":
            x = remove_small_objects_device(x, d.get("min_size", 64), d.get("connectivity", 1))
        elif name == "CastToType":
            x = x.to(torch.uint8)
        else:
            raise NotImplementedError(f"post-processing step {name} is not on the GPU path")
    return x
