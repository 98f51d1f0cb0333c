"""The reference's object API (Greenhouse / Forest / SimulationSpace, generate_vessel_graph.py:24-53): the constructors' draws from the
GLOBAL generators on CPU against the reference-made CSV fixtures; the device run behind develop_forest() in the gpu-marked test."""
import csv
import io
import os
import random

import numpy as np
import pytest
import yaml

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "sim_golden.npz")


def _case(g, name, tmp_path):
    seed, i1, i2 = (int(v) for v in g[name + "_seed_I"])
    cfg = yaml.safe_load(str(g["nerve_config_yaml" if name.startswith("nerve_") else "config_yaml"]))
    cfg["Greenhouse"]["modes"][0]["I"], cfg["Greenhouse"]["modes"][1]["I"] = i1, i2
    if name.startswith("geom_"):
        path = str(tmp_path / "geometry.npy")
        np.save(path, g["geometry_mask"])
        cfg["Greenhouse"]["SimulationSpace"]["oxygen_sample_geometry_path"] = path
    return seed, cfg


def _construct(cfg, seed):
    """generate_vessel_graph.py:24-37 after `random.seed(seed); np.random.seed(seed)`."""
    from octa_autosegmentation_amd.vessel_graph_generation.forest import Forest
    from octa_autosegmentation_amd.vessel_graph_generation.greenhouse import Greenhouse
    random.seed(seed)
    np.random.seed(seed)
    greenhouse = Greenhouse(cfg["Greenhouse"])
    kw = dict(nerve_center=greenhouse.nerve_center, nerve_radius=greenhouse.nerve_radius)
    art = Forest(cfg["Forest"], greenhouse.d, greenhouse.r, greenhouse.simspace, **kw)
    ven = Forest(cfg["Forest"], greenhouse.d, greenhouse.r, greenhouse.simspace, arterial=False, **kw)
    greenhouse.set_forests(art, ven)
    return greenhouse, art, ven


def _rows(forests):
    """The row list of generate_vessel_graph.py:41-53 written the way :59-66 writes it."""
    buf = io.StringIO()
    w = csv.writer(buf)
    w.writerow(["node1", "node2", "radius"])
    for forest in forests:
        for tree in forest.get_trees():
            for node in tree.get_tree_iterator(exclude_root=True, only_active=False):
                w.writerow([node.position, node.get_proximal_node().position, node.radius])
    return buf.getvalue()


@pytest.mark.parametrize("name", ["run_s1_30_20", "run_s5_10_5", "nerve_s3_12_6", "geom_s6_10_5"])
def test_constructors_draw_what_the_reference_draws(name, tmp_path):
    """FAZ radius and the node pair of every stump edge (the row a tree starts with) equal the reference's; no GPU involved."""
    g = np.load(GOLDEN)
    seed, cfg = _case(g, name, tmp_path)
    greenhouse, art, ven = _construct(cfg, seed)
    assert greenhouse.FAZ_radius == float(g[name + "_faz"])
    want = [row.rsplit(",", 1)[0] for row in g[name + "_csv"].tobytes().decode().split("\r\n")]      # the radius grows later
    stump_lines = _rows([art, ven]).split("\r\n")[1:-1]
    assert len(stump_lines) == 2 * cfg["Forest"]["N_trees"]
    at = 0
    for line in stump_lines:                      # in order, each as a row of the grown graph
        at = want.index(line.rsplit(",", 1)[0], at) + 1
    assert greenhouse.simspace.valid_voxels.shape[1] == 3
    root = art.get_trees()[0].root
    assert root.is_root and not root.is_leaf and root.get_distal_node().get_proximal_node() is root
    with pytest.raises(RuntimeError):
        root.get_proximal_node()


def test_constructors_on_other_masks_and_z_walls(tmp_path):
    """Stump draws on masks of other shapes and on the z walls (simulation_space.py:70-76, forest.py:153-181) against the
    reference-made fixtures; without a geometry file the z walls fail as they do in the reference."""
    from _sim_cases import mask_cases
    for name, cfg, seed, g in mask_cases(tmp_path):
        greenhouse, art, ven = _construct(cfg, seed)
        assert greenhouse.FAZ_radius == float(g[name + "_faz"])
        want = [row.rsplit(",", 1)[0] for row in g[name + "_csv"].tobytes().decode().split("\r\n")]
        at = 0
        for line in _rows([art, ven]).split("\r\n")[1:-1]:
            at = want.index(line.rsplit(",", 1)[0], at) + 1
    gold = np.load(GOLDEN)
    seed, cfg = _case(gold, "run_s5_10_5", tmp_path)
    cfg["Forest"]["source_walls"] = {"z0": True}
    with pytest.raises(AttributeError):
        _construct(cfg, seed)


def test_unknown_forest_type_is_rejected():
    from octa_autosegmentation_amd.vessel_graph_generation.forest import Forest
    with pytest.raises(NotImplementedError):
        Forest({"type": "grid", "N_trees": 1}, 0.1, 0.01, type("S", (), {"shape": np.ones(3)})())


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["run_s1_30_20", "run_s11_20_0", "nerve_s3_12_6", "geom_s6_10_5"])
def test_develop_forest_equals_reference_run(name, tmp_path, hip_lib_built):
    """The whole flow of generate_vessel_graph.main through the object API: CSV text, O2 / CO2 fields, and the NEXT draw of both
    global generators after develop_forest() equal the reference's."""
    g = np.load(GOLDEN)
    seed, cfg = _case(g, name, tmp_path)
    greenhouse, art, ven = _construct(cfg, seed)
    greenhouse.develop_forest()
    _check_developed(g, name, cfg, greenhouse, art, ven, tmp_path)


@pytest.mark.gpu
def test_develop_forest_on_other_masks_and_z_walls(tmp_path, hip_lib_built):
    from _sim_cases import mask_cases
    for name, cfg, seed, g in mask_cases(tmp_path):
        greenhouse, art, ven = _construct(cfg, seed)
        greenhouse.develop_forest()
        _check_developed(g, name, cfg, greenhouse, art, ven, tmp_path / name)


def _check_developed(g, name, cfg, greenhouse, art, ven, tmp_path):
    assert _rows([art, ven]).encode() == g[name + "_csv"].tobytes()
    assert sum(1 for _ in art.get_nodes()) - cfg["Forest"]["N_trees"] == int(g[name + "_n_art"])
    assert (greenhouse.oxys == g[name + "_oxy"]).all() and (greenhouse.co2s == g[name + "_co2"]).all()
    if name + "_next" in g:         # the fixed-geometry fixtures were recorded without it
        assert [random.random(), float(np.random.random_sample())] == list(g[name + "_next"])
    leaf = [n for n in ven.get_nodes() if n.is_leaf][0]
    assert leaf.proximal_num_segments >= 1 and leaf.get_proximal_radius() == leaf.radius
    # the reference's per-step statistics and its save_stats plots (greenhouse.py:72-76, 128-134, 401-441)
    tr = g[name + "_trace"]
    assert greenhouse.art_nodes_per_step == [0] + tr[:, 0].tolist() and greenhouse.oxys_per_step == [0] + tr[:, 1].tolist()
    assert greenhouse.ven_nodes_per_step == [0] + tr[:, 2].tolist() and greenhouse.co2_per_step == [0] + tr[:, 3].tolist()
    assert len(greenhouse.time_per_step) == len(tr)
    greenhouse.save_stats(str(tmp_path / "stats"))
    assert sorted(os.listdir(tmp_path / "stats")) == ["co2_distribution.png", "growth_over_time.png", "oxy_distribution.png", "time_per_step.png"]
