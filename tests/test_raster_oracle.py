"""CPU: pins the raster/dither oracle against fixtures produced by the reference itself
(tools/make_golden_raster.py: matplotlib Agg via the reference's rasterize_forest, Pillow
convert("1"), and the reference's shipped label PNGs)."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from oracle import octa_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_full_graphs(raster_golden):
    g = raster_golden
    for k in range(int(g["n_graphs"])):
        e = g[f"graph{k}_edges"]
        assert (octa_oracle.rasterize(e, [304, 304]) == g[f"graph{k}_img304"]).all()
        img = octa_oracle.rasterize(e, [1216, 1216])
        assert (img == g[f"graph{k}_img1216"]).all()
        f = octa_oracle.rasterize(e, [1216, 1216], min_radius=0.0033, max_radius=1)
        assert hashlib.sha256(f.tobytes()).hexdigest() == str(g[f"graph{k}_img1216_minr_sha256"])
        # label = Floyd-Steinberg of the raster == the PNG shipped in the reference's datasets/labels
        label = np.unpackbits(g[f"graph{k}_label_packed"])[: 1216 * 1216].reshape(1216, 1216) * 255
        assert (octa_oracle.fs_dither(img) == label).all()


def test_oracle_synthetic(raster_golden):
    g = raster_golden
    for t in range(int(g["n_syn"])):
        W, H, mip = (int(v) for v in g[f"syn{t}_res"])
        img = octa_oracle.rasterize(g[f"syn{t}_edges"], [W, H], mip)
        assert img.shape == g[f"syn{t}_img"].shape
        assert (img == g[f"syn{t}_img"]).all(), f"synthetic case {t}"


def test_oracle_fs_dither(raster_golden):
    g = raster_golden
    for t in range(int(g["n_fs"])):
        assert (octa_oracle.fs_dither(g[f"fs{t}_in"]) == g[f"fs{t}_out"]).all()


def test_oracle_edge_cases():
    # empty graph, fully-outside edge, zero-length edge, radius window
    assert octa_oracle.rasterize(np.zeros((0, 7)), [32, 24]).sum() == 0
    e = np.array([[2.0, 2.0, 0, 3.0, 3.0, 0, 0.01], [0.5, 0.5, 0, 0.5, 0.5, 0, 0.01]])
    assert octa_oracle.rasterize(e, [32, 24]).sum() == 0
    e = np.array([[0.2, 0.2, 0, 0.8, 0.8, 0, 0.01]])
    assert octa_oracle.rasterize(e, [32, 32]).sum() > 0
    assert octa_oracle.rasterize(e, [32, 32], min_radius=0.02, max_radius=1).sum() == 0
    keep = np.array([0], np.uint8)
    assert octa_oracle.rasterize(e, [32, 32], keep=keep).sum() == 0


def test_kernel_arithmetic_on_host(raster_golden):
    """The HIP kernel's closed-form cell evaluation (csrc/raster_core.h compiled as host C++)
    against the golden images: catches kernel-math regressions without a GPU."""
    import ctypes
    src = os.path.join(ROOT, "tests", "native", "raster_core_host.cpp")
    so = os.path.join(ROOT, "tests", "native", "libcorehost.so")
    hdr = os.path.join(ROOT, "octa_autosegmentation_amd", "csrc", "raster_core.h")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src])
    ch = ctypes.CDLL(so)
    ch.octa_corehost_rasterize.restype = ctypes.c_long
    ch.octa_corehost_rasterize.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p]

    def core(e, W, H, mip=2):
        e = np.ascontiguousarray(e, dtype=np.float64)
        out = np.zeros((H, W), np.uint8)
        n = ch.octa_corehost_rasterize(e.ctypes.data, len(e), W, H, mip, -np.inf, np.inf, None, out.ctypes.data)
        assert n >= 0
        return out

    ch.octa_corehost_rasterize_acc.restype = ctypes.c_long
    ch.octa_corehost_rasterize_acc.argtypes = ch.octa_corehost_rasterize.argtypes

    def core_acc(e, W, H, mip=2):
        """the item-parallel form of the fold (round 4): pieces walked cell by cell into per-block accumulators (hline_cells)"""
        e = np.ascontiguousarray(e, dtype=np.float64)
        out = np.zeros((H, W), np.uint8)
        n = ch.octa_corehost_rasterize_acc(e.ctypes.data, len(e), W, H, mip, -np.inf, np.inf, None, out.ctypes.data)
        assert n >= 0
        return out

    g = raster_golden
    for t in range(int(g["n_syn"])):
        W, H, mip = (int(v) for v in g[f"syn{t}_res"])
        assert (core(g[f"syn{t}_edges"], W, H, mip) == g[f"syn{t}_img"]).all(), f"synthetic case {t}"
        assert (core_acc(g[f"syn{t}_edges"], W, H, mip) == g[f"syn{t}_img"]).all(), f"synthetic case {t} (accumulator form)"
    assert (core(g["graph0_edges"], 304, 304) == g["graph0_img304"]).all()
    assert (core(g["graph1_edges"], 1216, 1216) == g["graph1_img1216"]).all()
    assert (core_acc(g["graph0_edges"], 304, 304) == g["graph0_img304"]).all()
    assert (core_acc(g["graph1_edges"], 1216, 1216) == g["graph1_img1216"]).all()


def test_oracle_reproduces_16_more_shipped_pairs():
    """tests/golden/raster_shipped_golden.npz (tools/make_golden_raster_shipped.py): 16 further csv <-> label pairs the reference ships
    (datasets/vessel_graphs/*.csv <-> datasets/labels/*.png), as data: edge array, the shipped label's bits, SHA-256 of the reference's own
    304 x 304 and 1216 x 1216 rasters. The oracle must reproduce all of them; with raster_golden.npz's two that is 18 shipped labels."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "raster_shipped_golden.npz"))
    names = [str(n) for n in g["names"]]
    assert len(names) >= 16
    for k, name in enumerate(names):
        e = g[f"edges_{k}"]
        assert hashlib.sha256(octa_oracle.rasterize(e, [304, 304]).tobytes()).hexdigest() == str(g[f"img304_sha256_{k}"]), name
        img = octa_oracle.rasterize(e, [1216, 1216])
        assert hashlib.sha256(img.tobytes()).hexdigest() == str(g[f"img1216_sha256_{k}"]), name
        label = np.unpackbits(g[f"label_packed_{k}"])[: 1216 * 1216].reshape(1216, 1216) * 255
        assert (octa_oracle.fs_dither(img) == label).all(), name


@pytest.mark.skipif(not os.path.isdir("/root/reference/datasets/labels"), reason="reference datasets not present")
def test_oracle_reproduces_shipped_labels_sample():
    """In the build container only: a spread of the 500 shipped csv<->label pairs."""
    import csv
    from PIL import Image
    names = sorted(os.listdir("/root/reference/datasets/vessel_graphs"))[::100]
    for fn in names:
        with open(os.path.join("/root/reference/datasets/vessel_graphs", fn), newline="") as fh:
            rows = list(csv.DictReader(fh))
        p = lambda s: [float(c) for c in s[1:-1].split(" ") if len(c) > 0]
        e = np.array([p(r["node1"]) + p(r["node2"]) + [float(r["radius"])] for r in rows])
        lab = np.array(Image.open(os.path.join("/root/reference/datasets/labels", fn[:-4] + ".png")).convert("L"))
        assert (octa_oracle.fs_dither(octa_oracle.rasterize(e, [1216, 1216])) == lab).all(), fn
