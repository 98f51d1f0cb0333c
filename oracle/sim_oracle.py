"""oracle/sim_oracle.py -- TEST INFRASTRUCTURE ONLY.

Python driver of the sequential CPU restatement of the vessel-graph simulator
(oracle/sim_oracle.cpp). Supplies the one piece that has to go through numpy/LAPACK -- the leaf
bifurcation geometry of greenhouse.py:205-233 (np.mean / np.cov / np.linalg.eig, the sign of the
dgeev eigenvector is part of the result) -- as a callback, and formats edges as the reference's
CSV text (generate_vessel_graph.py:59-66).
"""
import ctypes
import csv
import io

import numpy as np

from . import octa_oracle

MODE_KEYS = ["I", "N", "eps_n", "eps_s", "eps_k", "delta_art", "delta_ven", "gamma_art", "gamma_ven", "phi", "omega",
             "kappa", "delta_sigma"]


class SimParams(ctypes.Structure):
    _fields_ = [("param_scale", ctypes.c_double), ("d", ctypes.c_double), ("r", ctypes.c_double),
                ("faz_radius_mean", ctypes.c_double), ("faz_radius_std", ctypes.c_double),
                ("rotation_radius", ctypes.c_double), ("faz_center", ctypes.c_double * 2),
                ("size", ctypes.c_double * 3), ("n_trees", ctypes.c_int), ("walls", ctypes.c_int * 4),
                ("n_modes", ctypes.c_int), ("modes", (ctypes.c_double * 13) * 8),
                ("forest_type", ctypes.c_int), ("nerve_center", ctypes.c_double * 2), ("nerve_radius", ctypes.c_double),
                ("geometry", ctypes.c_void_p), ("geometry_shape", ctypes.c_int * 3),
                ("n_source_walls", ctypes.c_int), ("source_walls", ctypes.c_int * 6)]


class SimResult(ctypes.Structure):
    _fields_ = [(k, ctypes.c_long) for k in ["n_art_edges", "n_ven_edges", "n_iter", "np_u32_draws", "py_random_draws",
                                              "murray_steps", "n_bifurcations", "nn_queries", "ball_queries"]] + \
               [("faz_radius", ctypes.c_double)]


BIF_CB = ctypes.CFUNCTYPE(None, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.c_int,
                          ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.POINTER(ctypes.c_double))


def _unit(v):
    return v / np.linalg.norm(v)


def bifurcation_geometry(position, atts, r, kappa, d):
    """Positions of the two children of a bifurcating leaf (greenhouse.py:205-233): Murray angles
    from equal child radii, in-plane axis = dominant eigenvector of the attractor covariance."""
    r_p = (r ** kappa + r ** kappa) ** (1 / kappa)
    phi_1 = np.degrees(np.arccos((r_p ** 4 + r ** 4 - r ** 4) / (2 * r_p ** 2 * r ** 2)))
    phi_2 = phi_1
    c = np.mean(atts, axis=0)
    dpc = c - position
    if np.linalg.norm(dpc) != 0.0:
        dpc = dpc / np.linalg.norm(dpc)
    X = np.array([a - c for a in atts]).transpose()
    w, v = np.linalg.eig(np.cov(X))
    d_l = v[:, np.argmax(w)]
    p1 = np.real(position + _unit(np.cos(np.radians(phi_1)) * dpc + np.sin(np.radians(phi_1)) * d_l) * d)
    p2 = np.real(position + _unit(np.cos(np.radians(phi_2)) * dpc - np.sin(np.radians(phi_2)) * d_l) * d)
    return p1, p2


@BIF_CB
def _bif_cb(pos, atts, n, r, kappa, d, out6):
    position = np.array([pos[0], pos[1], pos[2]])
    A = np.array([[atts[3 * i], atts[3 * i + 1], atts[3 * i + 2]] for i in range(n)])
    p1, p2 = bifurcation_geometry(position, A, r, kappa, d)
    for i in range(3):
        out6[i] = float(p1[i])
        out6[3 + i] = float(p2[i])


_lib = None


def lib():
    global _lib
    if _lib is None:
        l = octa_oracle.lib()
        l.octa_oracle_simulate.restype = ctypes.c_int
        l.octa_oracle_simulate.argtypes = [ctypes.POINTER(SimParams), ctypes.c_uint32, ctypes.c_uint64, BIF_CB,
                                           ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_long,
                                           ctypes.c_void_p, ctypes.c_long, ctypes.POINTER(ctypes.c_long),
                                           ctypes.c_void_p, ctypes.c_long, ctypes.POINTER(ctypes.c_long),
                                           ctypes.POINTER(SimResult)]
        l.octa_oracle_hash_tuple3.restype = ctypes.c_uint64
        l.octa_oracle_hash_tuple3.argtypes = [ctypes.c_void_p]
        l.octa_oracle_set_order.restype = ctypes.c_long
        l.octa_oracle_set_order.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        l.octa_oracle_kd_indices.restype = None
        l.octa_oracle_kd_indices.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        l.octa_oracle_np_stream.restype = None
        l.octa_oracle_np_stream.argtypes = [ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                            ctypes.c_uint32, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        l.octa_oracle_py_stream.restype = None
        l.octa_oracle_py_stream.argtypes = [ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int,
                                            ctypes.c_void_p]
        _lib = l
    return _lib


def params_from_config(config):
    """config = the full generator YAML dict (keys Greenhouse / Forest / output)."""
    g, f = config["Greenhouse"], config["Forest"]
    if f["type"] not in ("stumps", "nerve"):
        raise NotImplementedError("oracle covers Forest.type 'stumps' (forest.py:68-181) and 'nerve' (forest.py:38-66)")
    p = SimParams()
    geo_path = g["SimulationSpace"].get("oxygen_sample_geometry_path")
    if geo_path is not None:      # simulation_space.py:29-34
        geo = np.ascontiguousarray(np.load(geo_path) != 0, dtype=np.uint8)
        p._geometry_keepalive = geo
        p.geometry = geo.ctypes.data
        p.geometry_shape[0], p.geometry_shape[1], p.geometry_shape[2] = geo.shape
    p.param_scale, p.d, p.r = g["param_scale"], g["d"], g["r"]
    p.faz_radius_mean, p.faz_radius_std = g["FAZ_radius_bound"]
    p.rotation_radius = g["rotation_radius"]
    p.faz_center[0], p.faz_center[1] = g["FAZ_center"]
    s = g["SimulationSpace"]
    p.size[0], p.size[1], p.size[2] = s["no_voxel_x"], s["no_voxel_y"], s["no_voxel_z"]
    p.forest_type = 1 if f["type"] == "nerve" else 0
    p.nerve_center[0], p.nerve_center[1] = g["nerve_center"]
    p.nerve_radius = g["nerve_radius"]
    p.n_trees = f["N_trees"]
    walls = f["source_walls"]
    names = ("x0", "x1", "y0", "y1", "z0", "z1")
    enabled = [k for k, v in walls.items() if v]          # forest.py:81-84: the order of the mapping
    if (walls.get("z0") or walls.get("z1")) and geo_path is None:
        raise AttributeError("'SimulationSpace' object has no attribute 'valid_pixels'")   # simulation_space.py:83
    for i, k in enumerate(names[:4]):
        p.walls[i] = 1 if walls.get(k) else 0
    p.n_source_walls = len(enabled)
    for i, k in enumerate(enabled):
        p.source_walls[i] = names.index(k)
    p.n_modes = len(g["modes"])
    for m, mode in enumerate(g["modes"]):
        for j, key in enumerate(MODE_KEYS):
            p.modes[m][j] = float(mode[key])
    return p


def simulate(config, seed, max_edges=40000, return_fields=False):
    """One sample with `random.seed(seed); np.random.seed(seed)` semantics. Returns edges [n,7]
    (arterial then venous, CSV row order) and an info dict."""
    p = params_from_config(config)
    edges = np.zeros((max_edges, 7))
    n_it = sum(int(m["I"]) for m in config["Greenhouse"]["modes"] if m["I"] > 0)
    trace = np.zeros((max(n_it, 1), 4), dtype=np.int64)
    res = SimResult()
    cap = 60000
    oxy, co2 = np.zeros((cap, 3)), np.zeros((cap, 3))
    n_oxy, n_co2 = ctypes.c_long(), ctypes.c_long()
    rc = lib().octa_oracle_simulate(ctypes.byref(p), int(seed) & 0xffffffff, int(abs(seed)), _bif_cb,
                                    edges.ctypes.data, max_edges, trace.ctypes.data, len(trace),
                                    oxy.ctypes.data, cap, ctypes.byref(n_oxy), co2.ctypes.data, cap,
                                    ctypes.byref(n_co2), ctypes.byref(res))
    if rc != 0:
        raise RuntimeError(f"octa_oracle_simulate failed: {rc}")
    n = res.n_art_edges + res.n_ven_edges
    info = {k: getattr(res, k) for k, _ in SimResult._fields_}
    info["trace"] = trace[: res.n_iter]
    if return_fields:
        info["oxy"] = oxy[: n_oxy.value].copy()
        info["co2"] = co2[: n_co2.value].copy()
    return edges[:n].copy(), info


def edges_to_csv_text(edges):
    """The reference's CSV bytes for an edge array (generate_vessel_graph.py:59-66)."""
    buf = io.StringIO(newline="")
    w = csv.writer(buf)
    w.writerow(["node1", "node2", "radius"])
    for e in edges:
        w.writerow([np.array(e[0:3]), np.array(e[3:6]), float(e[6])])
    return buf.getvalue()
