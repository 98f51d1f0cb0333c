"""Voxeliser: oracle vs reference-made fixtures (CPU) and HIP kernel vs oracle + fixtures (GPU)."""
import os

import numpy as np
import pytest

from oracle import octa_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def vg():
    return np.load(os.path.join(ROOT, "tests", "golden", "voxel_golden.npz"))


def test_oracle_matches_reference_fixtures(vg):
    for k in range(int(vg["n_cases"])):
        iz, mn, mx = vg[f"case{k}_args"]
        vol = octa_oracle.voxelize(vg[f"case{k}_edges"], vg[f"case{k}_dims"], mn, mx, None, bool(iz))
        assert vol.shape == vg[f"case{k}_vol"].shape
        assert (vol == vg[f"case{k}_vol"]).all(), k


@pytest.mark.gpu
def test_hip_voxeliser_matches_fixtures_and_oracle(vg, hip_lib_built):
    import torch
    from octa_autosegmentation_amd.vessel_graph_generation import tree2img
    for k in range(int(vg["n_cases"])):
        iz, mn, mx = vg[f"case{k}_args"]
        e = vg[f"case{k}_edges"]
        forest = [{"node1": e[i, 0:3].copy(), "node2": e[i, 3:6].copy(), "radius": e[i, 6]} for i in range(len(e))]
        rl = []
        vol, bd = tree2img.voxelize_forest(forest, [int(v) for v in vg[f"case{k}_dims"]], rl, min_radius=mn, max_radius=mx, ignore_z=bool(iz))
        assert vol.dtype == np.uint16 and (vol == vg[f"case{k}_vol"]).all(), k
        assert len(rl) == int(vg[f"case{k}_n_radius"])
    # ragged batch vs oracle, incl. an empty graph
    rng = np.random.default_rng(2)
    graphs = [vg["case0_edges"][:50], np.zeros((0, 7)), vg["case1_edges"], vg["case0_edges"][100:400]]
    off = np.zeros(len(graphs) + 1, np.int64); off[1:] = np.cumsum([len(x) for x in graphs])
    d = torch.from_numpy(np.concatenate(graphs)).cuda()
    out = tree2img.voxelize_edges_device(d, off, [128, 96, 8]).cpu().numpy().view(np.uint16)
    for b, gph in enumerate(graphs):
        assert (out[b] == octa_oracle.voxelize(gph, [128, 96, 8])).all(), b
