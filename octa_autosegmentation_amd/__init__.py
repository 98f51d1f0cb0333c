"""octa_autosegmentation_amd -- MI355X-native hot path of aiforvision/OCTA-autosegmentation.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all heavy
arithmetic runs in hand-written HIP kernels behind the C-ABI declared in include/octa_hip.h
(liboctahip.so, bound with ctypes in _native.py). There is no CPU fallback: every product
entry point raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"
