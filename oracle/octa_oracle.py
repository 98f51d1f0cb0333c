"""oracle/octa_oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes front-end of the CPU checker (oracle/*.c -> liboctaoracle.so). Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (octa_autosegmentation_amd/) never does.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboctaoracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".c", ".cpp", ".h")) or f == "Makefile"]
    stale = (not os.path.exists(LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if force or stale:
        res = subprocess.run(["make", "-C", HERE, "-B" if force else "-s"], stdout=subprocess.PIPE,
                             stderr=subprocess.STDOUT, text=True)
        if res.returncode != 0:
            raise RuntimeError("building the oracle failed:\n" + res.stdout)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        l = ctypes.CDLL(LIB_PATH)
        l.octa_oracle_rasterize.restype = ctypes.c_long
        l.octa_oracle_rasterize.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]
        l.octa_oracle_fs_dither.restype = None
        l.octa_oracle_fs_dither.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        l.octa_oracle_voxel_dims.restype = None
        l.octa_oracle_voxel_dims.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        l.octa_oracle_voxelize.restype = ctypes.c_long
        l.octa_oracle_voxelize.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_double, ctypes.c_double,
                                           ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        _lib = l
    return _lib


def rasterize(edges, image_resolution, MIP_axis=2, min_radius=-np.inf, max_radius=np.inf, keep=None):
    """edges float64 [n,7] -> uint8 [no_pixels_y, no_pixels_x] (tree2img.py:65-113 restated)."""
    edges = np.ascontiguousarray(edges, dtype=np.float64).reshape(-1, 7)
    W, H = int(image_resolution[0]), int(image_resolution[1])
    out = np.zeros((H, W), np.uint8)
    kp = None
    if keep is not None:
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
        kp = keep.ctypes.data
    n = lib().octa_oracle_rasterize(edges.ctypes.data, len(edges), W, H, int(MIP_axis), float(min_radius),
                                    float(max_radius), kp, out.ctypes.data)
    if n < 0:
        raise RuntimeError(f"octa_oracle_rasterize failed: {n}")
    return out


def fs_dither(img):
    """uint8 [H,W] -> uint8 {0,255} [H,W]; Pillow convert('1') restated (visualize_vessel_graphs.py:99)."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    H, W = img.shape
    out = np.zeros_like(img)
    lib().octa_oracle_fs_dither(img.ctypes.data, W, H, out.ctypes.data)
    return out


def voxelize(edges, volume_dimensions, min_radius=-np.inf, max_radius=np.inf, keep=None, ignore_z=False):
    """edges float64 [n,7] -> uint16 [X,Y,Z] with the reference's padded dims (tree2img.py:176-280 restated)."""
    edges = np.ascontiguousarray(edges, dtype=np.float64).reshape(-1, 7)
    dims = np.ascontiguousarray(volume_dimensions, dtype=np.int32)
    pd = np.zeros(3, np.int32)
    lib().octa_oracle_voxel_dims(dims.ctypes.data, pd.ctypes.data)
    out = np.zeros(tuple(int(v) for v in pd), np.uint16)
    kp = None
    if keep is not None:
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
        kp = keep.ctypes.data
    lib().octa_oracle_voxelize(edges.ctypes.data, len(edges), dims.ctypes.data, float(min_radius), float(max_radius), kp,
                               int(bool(ignore_z)), out.ctypes.data)
    return out
