"""Graph file formats of the reference: the `node1,node2,radius` CSV (generate_vessel_graph.py:59-66,
forest.py:196-207) and its read-back (visualize_vessel_graphs.py:71-75, data_transforms.py:369-375).

Positions are written with str(np.ndarray) -- numpy's default print options, 8 fractional digits,
fixed or (for the whole 3-vector) scientific notation -- so everything downstream of a CSV (the label
rasteriser in particular) sees positions rounded to that text. `positions_as_read_back` reproduces
that rounding arithmetically for whole arrays, so a batch can be rasterised on the GPU exactly as if
it had gone through the file, without formatting and re-parsing millions of numbers.
"""
import csv
import io

import numpy as np


def edges_to_csv_text_numpy(edges):
    """The CSV text produced the reference's way -- csv.writer over str(np.ndarray) and repr(float), as
    generate_vessel_graph.py:59-66 does: the yardstick of the native formatter (tests/test_fileio.py), 45x slower."""
    buf = io.StringIO(newline="")
    w = csv.writer(buf)
    w.writerow(["node1", "node2", "radius"])
    for e in edges:
        w.writerow([np.array(e[0:3]), np.array(e[3:6]), float(e[6])])
    return buf.getvalue()


def edges_to_csv_bytes(edges):
    """CSV bytes exactly as the reference writes them, formatted natively (csrc/fileio.cpp: numpy's str(ndarray) and
    CPython's repr(float) restated; byte-identical to edges_to_csv_text_numpy)."""
    import ctypes
    from . import _native
    e = np.ascontiguousarray(edges, dtype=np.float64).reshape(-1, 7)
    lib = _native.lib()
    cap = lib.octa_csv_bytes_bound(len(e))
    buf = ctypes.create_string_buffer(cap)
    n = lib.octa_csv_format_edges(e.ctypes.data, len(e), buf, cap)
    if n < 0:
        _native.check(int(n), "octa_csv_format_edges")
    return buf.raw[:n]


def edges_to_csv_text(edges):
    return edges_to_csv_bytes(edges).decode("ascii")


def write_csv(edges, path):
    """`<name>.csv` of generate_vessel_graph.py:59-66, formatted and written natively (the GIL is released for the call)."""
    from . import _native
    e = np.ascontiguousarray(edges, dtype=np.float64).reshape(-1, 7)
    _native.check(_native.lib().octa_csv_write_file(str(path).encode(), e.ctypes.data, len(e)), "octa_csv_write_file")


def parse_legacy_position(s):
    # tree2img.py:73-76
    return [float(c) for c in s[1:-1].split(" ") if len(c) > 0]


def read_csv(path):
    """CSV -> float64 [n,7] through Python's csv module (the positions as the reference's 'Legacy' string branch parses
    them): the yardstick of read_csv_native."""
    rows = []
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            rows.append(parse_legacy_position(row["node1"]) + parse_legacy_position(row["node2"]) + [float(row["radius"])])
    return np.asarray(rows, dtype=np.float64).reshape(-1, 7)


def parse_csv_bytes(text):
    from . import _native
    lib = _native.lib()
    n = lib.octa_csv_count_rows(text, len(text))
    out = np.zeros((max(int(n), 0), 7))
    got = lib.octa_csv_parse_edges(text, len(text), out.ctypes.data, len(out))
    if got < 0:
        _native.check(int(got), "octa_csv_parse_edges")
    return out[:got]


def read_csv_native(path):
    """CSV -> float64 [n,7] with the native reader (strtod per number, as float() does): identical to read_csv, 60x faster."""
    with open(path, "rb") as f:
        return parse_csv_bytes(f.read())


def _round_decimals(x, scale):
    """nearest double to round(x * scale) / scale for exactly representable powers of ten `scale`."""
    k = np.rint(np.asarray(x, dtype=np.longdouble) * np.asarray(scale, dtype=np.longdouble)).astype(np.float64)
    return k / scale


def positions_as_read_back(pos):
    """float64 [n,3] -> the values float(str(np.array(row))) would give (numpy 2.x default printing):
    scientific notation with 8 mantissa decimals when min|x| < 1e-4 or max|x|/min|x| > 1e3 over the
    non-zero entries of the row (or max >= 1e8), otherwise fixed notation with 8 decimals."""
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, 3)
    a = np.abs(pos)
    nz = np.where(a > 0, a, np.nan)
    with np.errstate(all="ignore"):
        mx = np.nanmax(nz, axis=1)
        mn = np.nanmin(nz, axis=1)
        sci = (mx >= 1e8) | (mn < 1e-4) | (mx / mn > 1000.0)
    sci = np.where(np.isnan(mx), False, sci)
    out = _round_decimals(pos, 1e8)
    if sci.any():
        p = pos[sci]
        ap = np.abs(p)
        with np.errstate(divide="ignore"):
            e = np.floor(np.log10(np.where(ap > 0, ap, 1.0))).astype(np.int64)
        # guard the log10 edge: mantissa must be in [1, 10)
        m = ap / np.power(10.0, e)
        e = np.where(m >= 10.0, e + 1, np.where((m < 1.0) & (ap > 0), e - 1, e))
        pw = 8 - e
        big = pw > 22  # not reachable for simulator coordinates; fall back to text for exactness
        scale = np.power(10.0, np.clip(pw, 0, 22).astype(np.float64))
        r = np.where(pw >= 0, _round_decimals(p, scale), p)
        if big.any() or (pw < 0).any():
            for i, j in zip(*np.nonzero(big | (pw < 0))):
                r[i, j] = float(np.format_float_scientific(p[i, j], precision=8, unique=True))
        out[sci] = np.where(ap > 0, r, p)
    return out


def edges_as_read_back(edges):
    e = np.array(edges, dtype=np.float64, copy=True).reshape(-1, 7)
    e[:, 0:3] = positions_as_read_back(e[:, 0:3])
    e[:, 3:6] = positions_as_read_back(e[:, 3:6])
    return e


def edges_as_read_back_device(d_edges):
    """CUDA float64 [n,7] -> the same edges with positions as they read back from the CSV text, computed on the
    GPU (octa_edges_read_back); rows outside the kernel's exact range (never produced by the simulator) fall
    back to the host formula."""
    import ctypes
    import torch
    from . import _native
    if d_edges.dtype != torch.float64 or not d_edges.is_cuda or not d_edges.is_contiguous():
        raise ValueError("d_edges must be a contiguous float64 CUDA tensor")
    out = torch.empty_like(d_edges)
    bad = ctypes.c_int(0)
    rc = _native.lib().octa_edges_read_back(_native.ctx(d_edges.device.index), ctypes.c_void_p(d_edges.data_ptr()),
                                            ctypes.c_void_p(out.data_ptr()), d_edges.shape[0], ctypes.byref(bad),
                                            _native.current_stream_ptr())
    _native.check(rc, "octa_edges_read_back")
    if bad.value:
        out = torch.from_numpy(edges_as_read_back(d_edges.cpu().numpy())).to(d_edges.device)
    return out
