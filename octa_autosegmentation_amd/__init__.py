"""octa_autosegmentation_amd -- MI355X-native hot path of aiforvision/OCTA-autosegmentation.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); all heavy
arithmetic runs in hand-written HIP kernels behind the C-ABI declared in include/octa_hip.h
(liboctahip.so, bound with ctypes in _native.py). There is no CPU fallback: every product
entry point raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"

import os as _os

# MIOpen serves the few layers that stay torch convolutions (7x7 stems of the generator, 4x4 stem / head of the PatchGAN, the
# fp32 reference modules). Its asm implicit-GEMM NHWC solvers are launched WITHOUT the workspace they ask for in immediate mode
# ("workspace required: 11214848, provided ptr: 0") and the data-gradient one faults (models/gan_seg_trainer.py): switched off
# for the whole process, before MIOpen reads its environment.
for _k in ("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_BWD_GTC_XDLOPS_NHWC", "MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC"):
    _os.environ.setdefault(_k, "0")
