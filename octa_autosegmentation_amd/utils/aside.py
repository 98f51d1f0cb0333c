"""Logging work of a training step on a side stream.

The post-transform of the scored sample (sigmoid, threshold, component labelling: reference models/lambda_model.py:48-52) and the
metric counts (utils/metrics.py) are ~35 short kernels that nothing in the step depends on; on the training stream they sit between
forward and backward (0.3 ms of an 18 ms DynUNet-S step). `aside(device, *tensors)` runs its body on a per-device side stream that
first waits for what the current stream has queued so far; `join_aside(device)` makes the current stream wait for the side stream
(before scores or plotted samples are read; `Metric.aggregate` does it for the scores). aside.ENABLED = False keeps everything on one stream.

The side stream has its OWN octa_ctx (round 4): a context's grow-only scratch (common.h: one context is used by one stream at a time) is
shared by the rasteriser and the component filter of the post-transform, and with an inline loader (`train.py --num_workers 0`) the
calling thread rasterises step k + 1 on the main stream while step k's component labelling may still be running on the side stream."""
import contextlib
import os

import torch

ENABLED = True       # module switch: False keeps the scored sample's post-transform and metric kernels on the training stream
_ASIDE = {}
_ASIDE_CTX = {}


def _aside_stream(device):
    device = torch.device(device)
    if device.type != "cuda" or not ENABLED:
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _ASIDE:
        _ASIDE[idx] = torch.cuda.Stream(device=idx)
    return _ASIDE[idx]


@contextlib.contextmanager
def aside(device, *tensors):
    s = _aside_stream(device)
    if s is None:
        yield
        return
    s.wait_stream(torch.cuda.current_stream(s.device))
    for t in tensors:
        if torch.is_tensor(t) and t.is_cuda:
            t.record_stream(s)             # the allocator must not hand the block out again before the side stream is done with it
    from .. import _native
    idx = s.device.index
    if idx not in _ASIDE_CTX:
        _ASIDE_CTX[idx] = _native.new_ctx(idx)          # lives as long as the process, like the stream
    with torch.cuda.stream(s), _native.use_ctx(_ASIDE_CTX[idx]):
        yield


def join_aside(device):
    s = _aside_stream(device)
    if s is not None:
        torch.cuda.current_stream(s.device).wait_stream(s)
