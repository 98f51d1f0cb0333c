"""CPU: the simulator's device code (csrc/sim_core.h, gpow.h, sim_host.h) compiled as host C++ with a
one-thread block (tests/native/sim_core_host.cpp) against the reference-made fixtures and the live
libraries it restates. Catches kernel-logic regressions without a GPU; not a product path."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import yaml

from oracle import sim_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def core():
    src = os.path.join(ROOT, "tests", "native", "sim_core_host.cpp")
    so = os.path.join(ROOT, "tests", "native", "libsimcorehost.so")
    deps = [src] + [os.path.join(ROOT, "octa_autosegmentation_amd", "csrc", f) for f in ("sim_core.h", "sim_host.h", "gpow.h", "glibc_pow_tables.h", "glibc_trig.h", "glibc_trig_tables.h")]
    if not os.path.exists(so) or max(os.path.getmtime(d) for d in deps) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src])
    l = ctypes.CDLL(so)
    l.octa_simcore_gpow_check.restype = ctypes.c_long
    l.octa_simcore_gpow_check.argtypes = [ctypes.c_long, ctypes.c_ulonglong]
    l.octa_simcore_gtrig.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    l.octa_simcore_kd_indices.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    l.octa_simcore_set_order.restype = ctypes.c_long
    l.octa_simcore_set_order.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    l.octa_simcore_host_run.restype = ctypes.c_int
    l.octa_simcore_host_run.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_ulonglong, sim_oracle.BIF_CB, ctypes.c_void_p,
                                        ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
    return l


def _load_core(so_name, extra_flags):
    src = os.path.join(ROOT, "tests", "native", "sim_core_host.cpp")
    so = os.path.join(ROOT, "tests", "native", so_name)
    deps = [src] + [os.path.join(ROOT, "octa_autosegmentation_amd", "csrc", f) for f in ("sim_core.h", "sim_host.h", "gpow.h", "glibc_pow_tables.h", "glibc_trig.h", "glibc_trig_tables.h")]
    if not os.path.exists(so) or max(os.path.getmtime(d) for d in deps) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-ffp-contract=off", "-fPIC", "-shared"] + extra_flags + ["-o", so, src])
    l = ctypes.CDLL(so)
    l.octa_simcore_host_run.restype = ctypes.c_int
    l.octa_simcore_host_run.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_ulonglong, sim_oracle.BIF_CB, ctypes.c_void_p,
                                        ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
    return l


@pytest.fixture(scope="module")
def core_large():
    """The same host harness compiled with -DOCTA_SIM_LARGE=1: the wide-field build of the phase code (32-bit indices, 64-bit kd
    elements, 18-bit packed indices)."""
    return _load_core("libsimcorehost_large.so", ["-DOCTA_SIM_LARGE=1"])


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "sim_golden.npz"))


def test_wide_field_build_reproduces_the_notebook_run(core_large):
    """f4 at full size, on the CPU: the reference's notebook configuration (example_custom_vessel_simulation.ipynb:138-156: 12 x 12 mm^2,
    optic-nerve forests with 16 trees, N = 8000, 400 + 500 iterations) through the wide-field build of the phase code, one host
    thread, against tests/golden/sim_f4_golden.npz -- the imported reference's own full-length run (tools/make_golden_sim_f4.py,
    26 minutes per seed): every one of the 900 trace rows and the SHA-256 of the 73 389-row CSV text. About 10 s."""
    import hashlib
    f4 = np.load(os.path.join(ROOT, "tests", "golden", "sim_f4_golden.npz"))
    cfg = yaml.safe_load(str(f4["config_yaml"]))
    seed = int(f4["seeds"][0])
    p = sim_oracle.params_from_config(cfg)
    n_it = 900
    cap = 200000
    edges = np.zeros((cap, 7)); trace = np.zeros((n_it, 4), np.int64); info = np.zeros(8, np.int64)
    rc = core_large.octa_simcore_host_run(ctypes.addressof(p), seed, seed, sim_oracle._bif_cb, edges.ctypes.data, cap, trace.ctypes.data, info.ctypes.data)
    assert rc == 0 and info[2] == 0 and info[7] == n_it
    assert (trace == f4[f"s{seed}_trace"]).all()
    assert info[0] == int(f4[f"s{seed}_rows"])
    assert hashlib.sha256(sim_oracle.edges_to_csv_text(edges[: info[0]]).encode()).hexdigest() == str(f4[f"s{seed}_csv_sha256"])



def test_gpow_is_glibc_pow(core):
    assert core.octa_simcore_gpow_check(3_000_000, 1) == 0
    assert core.octa_simcore_gpow_check(1_000_000, 987654321) == 0


def test_nth_element_restatement_matches_scipy(core):
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(12)
    for n in (1, 16, 17, 40, 300, 5000, 13000):
        pts = np.ascontiguousarray(rng.uniform(0, 1, (n, 3)) * np.array([1, 1, 0.0131]))
        out = np.zeros(n, np.uint16)
        core.octa_simcore_kd_indices(pts.ctypes.data, n, out.ctypes.data)
        assert (out == cKDTree(pts).indices).all(), n
    # round 3: the LDS elements are 32-bit words (18-bit quantised key << 14 | index); coordinates that share a quantisation bucket
    # are compared on the doubles themselves. Clustered data -- distinct doubles 1e-10 to 1e-6 apart around a handful of
    # centres -- sends most comparisons of the deep levels down that path
    for n, k in ((300, 3), (5000, 7), (13312, 40)):
        centres = rng.uniform(0.1, 0.9, (k, 3))
        pts = centres[rng.integers(0, k, n)] + rng.uniform(-1, 1, (n, 3)) * 10.0 ** rng.uniform(-10, -6, (n, 1))
        pts = np.ascontiguousarray(pts * np.array([1, 1, 0.0131]))
        assert len(np.unique(pts[:, 0])) == n and len(np.unique(pts[:, 1])) == n
        out = np.zeros(n, np.uint16)
        core.octa_simcore_kd_indices(pts.ctypes.data, n, out.ctypes.data)
        assert (out == cKDTree(pts).indices).all(), (n, k)


    # near-ties of the spreads: the single-precision outer boxes cannot order x against y, and the exact extrema of the range decide --
    # measured by the whole workgroup for ranges above 256 points (round 3). Spreads equal to the last bit (the first dimension
    # wins, as in scipy), y larger by one ulp, and a slab thin in x and y so that lower levels tie too
    for n, variant in ((5000, 0), (5000, 1), (9000, 2), (13000, 0)):
        pts = rng.uniform(0.05, 0.95, (n, 3)) * np.array([1, 1, 0.0131])
        pts[0, 0], pts[1, 0], pts[2, 1], pts[3, 1] = 0.0, 1.0, 0.0, 1.0                  # both spreads exactly 1
        if variant == 1:
            pts[3, 1] = np.nextafter(1.0, 2.0)                                           # y larger by one ulp: dimension 1 must win
        if variant == 2:
            pts[:, :2] = 0.5 + (pts[:, :2] - 0.5) * 1e-3                                 # spreads of 1e-3, equal to ~1e-19 relative
        pts = np.ascontiguousarray(pts)
        out = np.zeros(n, np.uint16)
        core.octa_simcore_kd_indices(pts.ctypes.data, n, out.ctypes.data)
        assert (out == cKDTree(pts).indices).all(), (n, variant)


def test_set_emulation_matches_cpython(core):
    rng = np.random.default_rng(5)
    pts = np.ascontiguousarray(rng.uniform(0, 1, (2500, 3)))
    for trial in range(25):
        n = int(rng.integers(1, 900))
        ids = rng.integers(0, n, int(n * 1.4)).astype(np.int32)
        tuples = [tuple(float(x) for x in pts[i]) for i in range(n)]
        s = set()
        for i in ids:
            s.add(tuples[i])
        want = [tuples.index(t) for t in s]
        out = np.zeros(n, np.int32)
        k = core.octa_simcore_set_order(pts.ctypes.data, ids.ctypes.data, len(ids), out.ctypes.data)
        assert out[:k].tolist() == want


def test_phases_reproduce_reference_csv(core, golden):
    for name in ("run_s0_30_20", "run_s3_30_20", "run_s5_10_5", "run_s11_20_0", "run_s4_0_12"):
        seed, i1, i2 = (int(v) for v in golden[name + "_seed_I"])
        cfg = yaml.safe_load(str(golden["config_yaml"]))
        cfg["Greenhouse"]["modes"][0]["I"] = i1
        cfg["Greenhouse"]["modes"][1]["I"] = i2
        p = sim_oracle.params_from_config(cfg)
        edges = np.zeros((40000, 7))
        trace = np.zeros((max(i1 + i2, 1), 4), np.int64)
        info = np.zeros(8, np.int64)
        rc = core.octa_simcore_host_run(ctypes.addressof(p), seed, seed, sim_oracle._bif_cb, edges.ctypes.data, 40000,
                                        trace.ctypes.data, info.ctypes.data)
        assert rc == 0 and info[2] == 0
        assert (trace[: info[7]] == golden[name + "_trace"]).all(), name
        text = sim_oracle.edges_to_csv_text(edges[: info[0]])
        assert text.encode() == golden[name + "_csv"].tobytes(), name
        # on the host (glibc's acos / cos / sin) the phase code equals the oracle in EVERY double, not only as printed
        e_or, _ = sim_oracle.simulate(cfg, seed)
        assert e_or.shape == (info[0], 7) and (e_or == edges[: info[0]]).all(), name


def test_assignment_bookkeeping_fallback_to_hbm_arrays(golden):
    """phase_assign keeps its group tables in the table area when they fit (sim_core.h); a forest too large for that uses the HBM
    arrays instead. -DOCTA_SIM_ASSIGN_FORCE_HBM takes that path always: same CSV bytes, same trace."""
    lib = _load_core("libsimcorehost_assign_hbm.so", ["-DOCTA_SIM_ASSIGN_FORCE_HBM"])
    for name in ("run_s0_30_20", "run_s5_10_5"):
        seed, i1, i2 = (int(v) for v in golden[name + "_seed_I"])
        cfg = yaml.safe_load(str(golden["config_yaml"]))
        cfg["Greenhouse"]["modes"][0]["I"] = i1
        cfg["Greenhouse"]["modes"][1]["I"] = i2
        p = sim_oracle.params_from_config(cfg)
        edges = np.zeros((40000, 7))
        trace = np.zeros((max(i1 + i2, 1), 4), np.int64)
        info = np.zeros(8, np.int64)
        rc = lib.octa_simcore_host_run(ctypes.addressof(p), seed, seed, sim_oracle._bif_cb, edges.ctypes.data, 40000,
                                       trace.ctypes.data, info.ctypes.data)
        assert rc == 0 and info[2] == 0
        assert (trace[: info[7]] == golden[name + "_trace"]).all(), name
        assert sim_oracle.edges_to_csv_text(edges[: info[0]]).encode() == golden[name + "_csv"].tobytes(), name


def test_phases_reproduce_reference_csv_on_other_masks(core, tmp_path):
    """The device phase code (host build) on the reference-made fixtures for other mask shapes and the z source walls
    (tests/golden/sim_masks_golden.npz; simulation_space.py:29-34, 70-76, forest.py:153-181)."""
    from _sim_cases import mask_cases
    n = 0
    for name, cfg, seed, g in mask_cases(tmp_path):
        p = sim_oracle.params_from_config(cfg)
        edges = np.zeros((40000, 7))
        trace = np.zeros((len(g[name + "_trace"]) + 1, 4), np.int64)
        info = np.zeros(8, np.int64)
        rc = core.octa_simcore_host_run(ctypes.addressof(p), seed, seed, sim_oracle._bif_cb, edges.ctypes.data, 40000,
                                        trace.ctypes.data, info.ctypes.data)
        assert rc == 0 and info[2] == 0
        assert (trace[: info[7]] == g[name + "_trace"]).all(), name
        assert sim_oracle.edges_to_csv_text(edges[: info[0]]).encode() == g[name + "_csv"].tobytes(), name
        n += 1
    assert n >= 4


def test_gtrig_is_glibc_sin_cos_acos(core):
    """csrc/glibc_trig.h (the acos / sin / cos that reach node positions) against this image's libm, every branch: sin / cos on
    [0, 2.4), acos on (0, 1] incl. arguments next to 1 and next to 0 and the interval knots."""
    import math
    rng = np.random.default_rng(7)
    n = 600000
    x = np.ascontiguousarray(np.concatenate([rng.uniform(0, 2.4, n - 6), [0.0, 1e-9, 0.126, 0.855469, math.pi / 2, 2.39]]))
    c = np.ascontiguousarray(np.concatenate([rng.uniform(0, 1, n // 2), 1 - 10 ** rng.uniform(-16, -1, n // 4), 10 ** rng.uniform(-18, 0, n // 4 - 6),
                                             [1.0, 0.125, 0.25, 0.5, 0.75, 0.96875]]))
    out = np.zeros((n, 3))
    core.octa_simcore_gtrig(x.ctypes.data, c.ctypes.data, n, out.ctypes.data)
    assert (out[:, 0] == np.array([math.sin(v) for v in x])).all()
    assert (out[:, 1] == np.array([math.cos(v) for v in x])).all()
    assert (out[:, 2] == np.array([math.acos(v) for v in c])).all()


@pytest.mark.parametrize("large", [False, True])
def test_phase_code_is_clean_under_address_and_ub_sanitizers(tmp_path, golden, large):
    """The simulator's phase code (csrc/sim_core.h) compiled for the host with -fsanitize=address,undefined and run on a short seeded
    simulation in a child process (libasan must be the first library of the process): no report, and the CSV text still equals the
    reference's. GPU AddressSanitizer is not available on the MI355X pool; this host build found two out-of-bounds writes of the
    wide-field build in round 3 and is the standing recipe for that class of defect."""
    import sys
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not installed")
    so = str(tmp_path / ("libsimcorehost_asan_large.so" if large else "libsimcorehost_asan.so"))
    src = os.path.join(ROOT, "tests", "native", "sim_core_host.cpp")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-mfma", "-ffp-contract=off", "-fPIC", "-shared", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"] + (["-DOCTA_SIM_LARGE=1"] if large else []) + ["-o", so, src])
    name = "run_s0_30_20"
    child = f'''
import ctypes, sys, hashlib
import numpy as np, yaml
sys.path.insert(0, {ROOT!r})
from oracle import sim_oracle
g = np.load({os.path.join(ROOT, "tests", "golden", "sim_golden.npz")!r})
cfg = yaml.safe_load(str(g["config_yaml"]))
cfg["Greenhouse"]["modes"][0]["I"], cfg["Greenhouse"]["modes"][1]["I"] = 30, 20
l = ctypes.CDLL({so!r})
l.octa_simcore_host_run.restype = ctypes.c_int
l.octa_simcore_host_run.argtypes = [ctypes.c_void_p, ctypes.c_uint, ctypes.c_ulonglong, sim_oracle.BIF_CB, ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p, ctypes.c_void_p]
p = sim_oracle.params_from_config(cfg)
edges = np.zeros((40000, 7)); trace = np.zeros((50, 4), np.int64); info = np.zeros(8, np.int64)
rc = l.octa_simcore_host_run(ctypes.addressof(p), 0, 0, sim_oracle._bif_cb, edges.ctypes.data, 40000, trace.ctypes.data, info.ctypes.data)
assert rc == 0 and info[2] == 0, (rc, info)
text = sim_oracle.edges_to_csv_text(edges[: info[0]])
assert text.encode() == g["{name}_csv"].tobytes()
print("SANITIZED-RUN-OK", info[0])
'''
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    r = subprocess.run([sys.executable, "-c", child], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "SANITIZED-RUN-OK" in r.stdout and "ERROR: AddressSanitizer" not in r.stdout and "runtime error" not in r.stdout, r.stdout[-3000:]
