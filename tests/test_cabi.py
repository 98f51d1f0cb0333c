"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/octa_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "octa_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(octa_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hip_lib_built):
    lib = ctypes.CDLL(hip_lib_built)
    syms = declared_symbols()
    assert len(syms) >= 8
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/octa_hip.h but not exported"


def test_python_binding_covers_header(hip_lib_built):
    from octa_autosegmentation_amd import _native
    assert sorted(_native.SIGNATURES) == declared_symbols()
    l = _native.lib()
    assert l.octa_abi_version() == 1


def test_no_gpu_is_loud(hip_lib_built):
    import pytest
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from octa_autosegmentation_amd import _native
    from octa_autosegmentation_amd.vessel_graph_generation import tree2img
    import numpy as np
    with pytest.raises(_native.OctaHipError):
        tree2img.rasterize_forest([{"node1": np.zeros(3), "node2": np.ones(3), "radius": 0.01}], [16, 16])


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "octa_autosegmentation_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                t = open(os.path.join(d, f)).read()
                assert "import oracle" not in t and "from oracle" not in t and "octaoracle" not in t, f
