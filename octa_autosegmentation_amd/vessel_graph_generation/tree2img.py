"""Drop-in for the reference's vessel_graph_generation/tree2img.py (2-D path), running on MI355X.

`rasterize_forest` keeps the reference signature, return types and side effects
(tree2img.py:12-114): edges are filtered by radius, optionally dropped (Python `random` stream,
blackdict bookkeeping) on the host exactly in the reference's order, and the surviving edges are
rendered by the HIP kernel behind `octa_rasterize_2d` -- bit-exact with what matplotlib's Agg
backend draws for the reference. There is no CPU path here.
"""
from random import random
from typing import Sequence, Tuple

import ctypes
import numpy as np

from .. import _native


def _parse_legacy(s: str):
    # tree2img.py:73-76 ("Legacy" string positions as read back from the CSV)
    return tuple([float(coord) for coord in s[1:-1].split(" ") if len(coord) > 0])


def select_edges(forest, min_radius=0, max_radius=1, max_dropout_prob=0, blackdict=None, radius_list=None,
                 radius_factor=1.3):
    """Host-side edge selection of tree2img.py:58-86 (also used by voxelize_forest, :212-235).

    Returns (edges float64 [n,7] of the edges to draw, blackdict). Consumes the global Python
    `random` stream exactly like the reference: one draw when no blackdict is passed, then one
    draw per in-range edge whose parent is not already black-listed.
    """
    if radius_list is None:
        radius_list = []
    if blackdict is None:
        blackdict = dict()
        p = random() ** 10 * max_dropout_prob
    else:
        p = 0
    rows = []
    for edge in forest:
        radius = float(edge["radius"])
        if radius < min_radius or radius > max_radius:
            continue
        n1 = edge["node1"]
        if isinstance(n1, np.ndarray) or isinstance(n1, list):
            current_node = tuple(n1)
            proximal_node = tuple(edge["node2"])
        elif isinstance(n1, str):
            current_node = _parse_legacy(n1)
            proximal_node = _parse_legacy(edge["node2"])
        else:
            raise TypeError(f"edge['node1'] must be ndarray, list or str, got {type(n1).__name__}")
        if proximal_node in blackdict or random() < p:
            blackdict[current_node] = True
            continue
        radius_list.append(radius * radius_factor)
        rows.append((*current_node, *proximal_node, radius))
    edges = np.asarray(rows, dtype=np.float64).reshape(-1, 7)
    return edges, blackdict


def rasterize_edges_device_plan(d_edges, edge_off, image_resolution, MIP_axis=2, min_radius=0.0, max_radius=1.0, d_keep=None):
    """First half of rasterize_edges_device (octa_rasterize_2d_plan): the per-edge records and the side-offset scan, with the
    rasteriser's one host synchronisation. Returns the token rasterize_edges_device_draw takes; the context's scratch holds ONE plan,
    so plan and draw of a call pair must not be interleaved with another pair on the same context (_native.use_ctx)."""
    import torch
    no_pixels_x, no_pixels_y = int(image_resolution[0]), int(image_resolution[1])
    edge_off = np.ascontiguousarray(edge_off, dtype=np.int64)
    B = len(edge_off) - 1
    if d_edges.dtype != torch.float64 or not d_edges.is_cuda or not d_edges.is_contiguous():
        raise ValueError("d_edges must be a contiguous float64 CUDA tensor")
    if d_edges.numel() != int(edge_off[-1]) * 7:
        raise ValueError("edge_off[-1] does not match the number of edges")
    h = _native.ctx(d_edges.device.index)
    rc = _native.lib().octa_rasterize_2d_plan(
        h, B, ctypes.c_void_p(d_edges.data_ptr()), ctypes.c_void_p(edge_off.ctypes.data),
        ctypes.c_void_p(d_keep.data_ptr()) if d_keep is not None else None,
        no_pixels_x, no_pixels_y, int(MIP_axis), float(min_radius), float(max_radius), _native.current_stream_ptr())
    _native.check(rc, "octa_rasterize_2d_plan")
    return dict(ctx=h, shape=(B, no_pixels_y, no_pixels_x), device=d_edges.device)


def rasterize_edges_device_draw(plan, out=None):
    """Second half (octa_rasterize_2d_draw): tessellation, row binning and the ordered fold of the planned call, enqueued on the
    current stream without a host synchronisation. Returns the uint8 CUDA tensor [B, no_pixels_y, no_pixels_x]."""
    import torch
    if out is None:
        out = torch.empty(plan["shape"], dtype=torch.uint8, device=plan["device"])
    if plan["shape"][0] > 0:
        rc = _native.lib().octa_rasterize_2d_draw(plan["ctx"], ctypes.c_void_p(out.data_ptr()), _native.current_stream_ptr())
        _native.check(rc, "octa_rasterize_2d_draw")
    return out


def rasterize_edges_device(d_edges, edge_off, image_resolution, MIP_axis=2, min_radius=0.0, max_radius=1.0,
                           d_keep=None, out=None):
    """Batched device entry: d_edges float64 CUDA tensor [n_total,7], edge_off int64 host array [B+1].

    Returns a uint8 CUDA tensor [B, no_pixels_y, no_pixels_x]. Asynchronous on the current stream.
    """
    return rasterize_edges_device_draw(rasterize_edges_device_plan(d_edges, edge_off, image_resolution, MIP_axis, min_radius, max_radius, d_keep), out)


def _rasterize_colorized_cpu(edges, image_resolution, MIP_axis, colorize):
    """float32 [H, W, 3] in 0..255: white-on-black geometry of the grey image, every edge in the plasma colour of its radius
    (`continous`: radius / 0.03 clipped to 1; `dicrete`: three classes at 0.01 / 0.02 -- the reference's spellings)."""
    if colorize not in ("continous", "dicrete"):
        raise NotImplementedError("Colorize only supports the options 'continous' or 'discrete'!")      # the reference's message
    try:
        import matplotlib
        matplotlib.use("Agg", force=False)
        from matplotlib import cm, collections
        from matplotlib import pyplot as plt
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("rasterize_forest(colorize=...) draws on the CPU with matplotlib, which is not installed") from e
    axes = [a for a in (0, 1, 2) if a != MIP_axis]
    no_pixels_x, no_pixels_y = image_resolution
    scale = max(no_pixels_x, no_pixels_y)
    widths = edges[:, 6] * 1.3 * scale
    c = widths / no_pixels_x / 1.3 * 3
    if colorize == "continous":
        c = np.minimum(c / 0.03, 1)
    else:
        c = np.where(c <= 0.01, 0.1, np.where(c <= 0.02, 0.5, 1.0))
    segs = [[(e[axes[1]], e[axes[0]]), (e[3 + axes[1]], e[3 + axes[0]])] for e in edges]
    fig = plt.figure(figsize=(no_pixels_x / 100, no_pixels_y / 100), dpi=100)
    try:
        fig.patch.set_facecolor("black")
        ax = plt.axes([0., 0., 1., 1.], frameon=False, xticks=[], yticks=[])
        ax.invert_yaxis()
        ax.add_collection(collections.LineCollection(segs, linewidths=widths, colors=cm.plasma(c), antialiaseds=True, capstyle="round"))
        fig.canvas.draw()
        data = np.frombuffer(fig.canvas.buffer_rgba(), dtype=np.uint8)
        img = data.reshape(fig.canvas.get_width_height()[::-1] + (4,))[:, :, :3]
        return np.array(img.astype(np.float32))
    finally:
        plt.close(fig)


def rasterize_forest(forest, image_resolution: Sequence[float], MIP_axis: int = 2, radius_list: list = None,
                     min_radius: float = 0, max_radius: float = 1, max_dropout_prob=0, blackdict=None,
                     colorize: str = None) -> Tuple[np.ndarray, dict]:
    """Same contract as the reference's rasterize_forest (tree2img.py:12-114): returns
    (uint16 [no_pixels_y, no_pixels_x] grey image in 0..255, blackdict)."""
    import torch
    edges, blackdict = select_edges(forest, min_radius, max_radius, max_dropout_prob, blackdict, radius_list)
    if colorize is not None:
        # visualisation option (radius-coloured RGB, tree2img.py:87-113), off the hot path: drawn on the CPU by matplotlib's Agg, the
        # reference's own rasteriser (SURVEY.md 8b allows the CPU here); edge selection, random draws and side effects are the same
        return _rasterize_colorized_cpu(edges, image_resolution, MIP_axis, colorize), blackdict
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    if dev is None:
        _native.ctx()  # raises: no CPU fallback
    d_edges = torch.from_numpy(edges).to(dev)
    # the radius window was applied on the host (it gates the RNG draws); pass the widest window
    img = rasterize_edges_device(d_edges, np.array([0, len(edges)]), image_resolution, MIP_axis,
                                 -np.inf, np.inf)
    return img[0].cpu().numpy().astype(np.uint16), blackdict


def voxelize_edges_device(d_edges, edge_off, volume_dimensions, min_radius=-np.inf, max_radius=np.inf, d_keep=None,
                          ignore_z=False):
    """Batched device entry of the 3-D voxeliser: uint16 CUDA tensor [B, X', Y', Z'] (padded dims)."""
    import torch
    edge_off = np.ascontiguousarray(edge_off, dtype=np.int64)
    B = len(edge_off) - 1
    dims = np.ascontiguousarray([int(v) for v in volume_dimensions], dtype=np.int32)
    pd = np.zeros(3, np.int32)
    _native.lib().octa_voxel_padded_dims(dims.ctypes.data, pd.ctypes.data)
    if d_edges.dtype != torch.float64 or not d_edges.is_cuda or not d_edges.is_contiguous():
        raise ValueError("d_edges must be a contiguous float64 CUDA tensor")
    out = torch.empty((B, int(pd[0]), int(pd[1]), int(pd[2])), dtype=torch.int16, device=d_edges.device)
    h = _native.ctx(d_edges.device.index)
    rc = _native.lib().octa_voxelize_3d(
        h, B, ctypes.c_void_p(d_edges.data_ptr()), ctypes.c_void_p(edge_off.ctypes.data),
        ctypes.c_void_p(d_keep.data_ptr()) if d_keep is not None else None, ctypes.c_void_p(dims.ctypes.data),
        float(min_radius), float(max_radius), int(bool(ignore_z)), ctypes.c_void_p(out.data_ptr()),
        _native.current_stream_ptr())
    _native.check(rc, "octa_voxelize_3d")
    return out


def voxelize_forest(forest, volume_dimensions: Sequence[float], radius_list: list = None, min_radius=0, max_radius=1,
                    max_dropout_prob=0, blackdict=None, ignore_z=False) -> Tuple[np.ndarray, dict]:
    """Same contract as the reference's voxelize_forest (tree2img.py:176-280): (uint16 [X',Y',Z'], blackdict);
    radius_list receives the unscaled radii (tree2img.py:235)."""
    import torch
    edges, blackdict = select_edges(forest, min_radius, max_radius, max_dropout_prob, blackdict, radius_list, radius_factor=1.0)
    if not torch.cuda.is_available():
        _native.ctx()
    d_edges = torch.from_numpy(edges).to(torch.device("cuda", torch.cuda.current_device()))
    vol = voxelize_edges_device(d_edges, np.array([0, len(edges)]), volume_dimensions, -np.inf, np.inf, None, ignore_z)
    return vol[0].cpu().numpy().view(np.uint16), blackdict


def binarize_label_device(d_img):
    """Pillow `convert("1")` (Floyd-Steinberg) of uint8 CUDA images [B,H,W] -> uint8 {0,255}.
    Mirrors visualize_vessel_graphs.py:97-99."""
    import torch
    if d_img.dtype != torch.uint8 or not d_img.is_cuda or not d_img.is_contiguous() or d_img.dim() != 3:
        raise ValueError("d_img must be a contiguous uint8 CUDA tensor [B,H,W]")
    B, H, W = d_img.shape
    out = torch.empty_like(d_img)
    h = _native.ctx(d_img.device.index)
    rc = _native.lib().octa_fs_dither(h, B, ctypes.c_void_p(d_img.data_ptr()), W, H,
                                      ctypes.c_void_p(out.data_ptr()), _native.current_stream_ptr())
    _native.check(rc, "octa_fs_dither")
    return out


def pack_label_bits_device(d_bits, out=None):
    """Mode "1" rows of binarised labels on the device (csrc/graphio.hip octa_pack_bits): uint8 CUDA [B,H,W] (non-zero = white) ->
    uint8 CUDA [B,H,(W+7)//8], bit 7 of a byte = its first pixel -- the rows a 1-bit PNG stores (visualize_vessel_graphs.py:99)."""
    import torch
    if d_bits.dtype != torch.uint8 or not d_bits.is_cuda or not d_bits.is_contiguous() or d_bits.dim() != 3:
        raise ValueError("d_bits must be a contiguous uint8 CUDA tensor [B,H,W]")
    B, H, W = d_bits.shape
    if out is None:
        out = torch.empty((B, H, (W + 7) // 8), dtype=torch.uint8, device=d_bits.device)
    rc = _native.lib().octa_pack_bits(_native.ctx(d_bits.device.index), ctypes.c_void_p(d_bits.data_ptr()), ctypes.c_void_p(out.data_ptr()), B * H, W,
                                      _native.current_stream_ptr())
    _native.check(rc, "octa_pack_bits")
    return out


def maximum_u8_device(a, b):
    """np.maximum(art_mat, ven_mat) of generate_vessel_graph.py:83 on device."""
    import torch
    if a.shape != b.shape or a.dtype != torch.uint8 or b.dtype != torch.uint8:
        raise ValueError("inputs must be uint8 tensors of equal shape")
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty_like(a)
    h = _native.ctx(a.device.index)
    rc = _native.lib().octa_max_u8(h, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                   ctypes.c_void_p(out.data_ptr()), a.numel(), _native.current_stream_ptr())
    _native.check(rc, "octa_max_u8")
    return out


def save_2d_img(img: np.ndarray, out_dir: str, name: str):
    """tree2img.py:282-292: `<out_dir>/<name>.png`, 8-bit grey, written by the native PNG encoder (csrc/fileio.cpp;
    the pixels decode identically to Pillow's file)."""
    a = np.ascontiguousarray(img.astype(np.uint8))
    if a.ndim != 2:
        raise ValueError("save_2d_img expects a 2-D image")
    _native.check(_native.lib().octa_png_write_gray8(f'{out_dir}/{name}.png'.encode(), a.ctypes.data, a.shape[1], a.shape[0], -1),
                  "octa_png_write_gray8")


def save_label_png(bits: np.ndarray, path: str):
    """visualize_vessel_graphs.py:99: a mode "1" PNG (non-zero = white) from a binarised label."""
    a = np.ascontiguousarray(bits.astype(np.uint8))
    _native.check(_native.lib().octa_png_write_bits(str(path).encode(), a.ctypes.data, a.shape[1], a.shape[0], -1), "octa_png_write_bits")
