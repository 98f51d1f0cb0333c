"""Seams of SURVEY.md 8(b) that sit beside the hot path: the reference's element containers (element_mesh.py:11-232) and the
`colorize=` option of rasterize_forest (tree2img.py:87-113)."""
import os
import random

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_colorize_runs_on_the_cpu_and_equals_the_reference():
    """tests/golden/colorize_golden.npz (tools/make_golden_colorize.py: the reference's own rasterize_forest with colorize=..., a dropout
    probability and a seeded `random`): same RGB pixels, same consumption of the global `random` stream, same blackdict size."""
    from octa_autosegmentation_amd.vessel_graph_generation import tree2img
    g = np.load(os.path.join(ROOT, "tests", "golden", "colorize_golden.npz"))
    forest = [{"node1": g["a"][i], "node2": g["b"][i], "radius": float(g["rad"][i])} for i in range(len(g["rad"]))]
    for mode in ("continous", "dicrete"):
        random.seed(3)
        radii = []
        img, black = tree2img.rasterize_forest(forest, [96, 80], 2, radius_list=radii, colorize=mode, max_dropout_prob=0.3)
        assert img.dtype == np.float32 and img.shape == (80, 96, 3)
        assert (img.astype(np.uint8) == g[mode]).all(), mode
        assert len(radii) + len(black) == len(forest)
    with pytest.raises(NotImplementedError):
        tree2img.rasterize_forest(forest, [96, 80], 2, colorize="rainbow")


def test_element_containers_small_sets_match_scipy():
    """One-leaf sets (<= 16 points) need no device: ball results in insertion order, nearest element, deletes by value / identity."""
    from scipy.spatial import cKDTree
    from octa_autosegmentation_amd.vessel_graph_generation.element_mesh import CoordKdTree, NodeKdTree, SpacePartitioner
    rng = np.random.default_rng(0)
    pts = [tuple(p) for p in rng.uniform(0, 1, (14, 3))]
    m = CoordKdTree()
    assert isinstance(m, SpacePartitioner) and m.find_nearest_element((0, 0, 0)) is None and m.find_elements_in_distance((0, 0, 0), 1) == []
    m.extend(pts)
    t = cKDTree(np.array(pts))
    for q in rng.uniform(0, 1, (20, 3)):
        assert m.find_elements_in_distance(q, 0.4) == [pts[i] for i in t.query_ball_point(q, 0.4)]
        assert m.find_nearest_element(q) == pts[t.query(q)[1]]
        d = t.query(q)[0]
        assert m.find_nearest_element(q, d * 0.999) is None and m.find_nearest_element(q, d) == pts[t.query(q)[1]]
    m.delete(pts[3]); m.delete((9, 9, 9)); m.delete_all([pts[5], pts[5], pts[7]])
    assert m.get_all_elements() == [p for i, p in enumerate(pts) if i not in (3, 5, 7)]
    m.add((0.5, 0.5, 0.5))
    assert m.find_nearest_element((0.5, 0.5, 0.51)) == (0.5, 0.5, 0.5)

    class N:
        def __init__(self, p):
            self.position = np.array(p)
    nodes = [N(p) for p in pts]
    nm = NodeKdTree()
    nm.extend(nodes)
    assert nm.find_nearest_element(pts[4]) is nodes[4]
    nm.delete(nodes[4])
    assert nm.find_nearest_element(pts[4]) is not nodes[4] and len(nm.get_all_elements()) == 13
    assert nm.find_nearest_elements([pts[0], pts[1]]) == [nodes[0], nodes[1]]


@pytest.mark.gpu
def test_element_container_ball_order_is_scipys_on_large_sets(hip_lib_built):
    """More than one leaf: the ball is reported in the order of scipy's tree.indices, which comes from the device's kd order
    (the code path of the simulator's O2 -> CO2 step)."""
    from scipy.spatial import cKDTree
    from octa_autosegmentation_amd.vessel_graph_generation.element_mesh import CoordKdTree
    rng = np.random.default_rng(5)
    pts = [tuple(p) for p in rng.uniform(0, 1, (3000, 3)) * np.array([1, 1, 0.013])]
    m = CoordKdTree()
    m.extend(pts)
    t = cKDTree(np.array(pts))
    qs = rng.uniform(0, 1, (40, 3)) * np.array([1, 1, 0.013])
    for q, got in zip(qs, m.find_elements_in_distances(qs, 0.05)):
        assert got == [pts[i] for i in t.query_ball_point(q, 0.05)]
    m.delete_all(pts[:500])                      # mutation: the index is rebuilt
    t = cKDTree(np.array(pts[500:]))
    assert m.find_elements_in_distance(qs[0], 0.06) == [pts[500 + i] for i in t.query_ball_point(qs[0], 0.06)]
