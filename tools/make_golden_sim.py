"""tools/make_golden_sim.py -- generates tests/golden/sim_golden.npz.

Runs ONLY in the build container: imports the reference's vessel_graph_generation package from
/root/reference (read-only; `anytree`, absent from the image, is replaced by the stand-in in
tools/ref_stubs/) and records, for seeded runs (`random.seed(s); np.random.seed(s)` before
Greenhouse(...) -- the reference itself never seeds), the CSV text the reference's export block
(generate_vessel_graph.py:43-66) produces, the per-iteration element counts
(greenhouse.py:129-134) and the final oxygen / CO2 fields. Fixtures are data only.

  python tools/make_golden_sim.py            # short runs (about a minute)
  python tools/make_golden_sim.py --full     # adds full-length (I=100+150) runs, ~70 s each
"""
import copy
import csv
import hashlib
import io
import os
import random
import sys

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "ref_stubs"))
sys.path.insert(0, "/root/reference")
from vessel_graph_generation.forest import Forest  # noqa: E402
from vessel_graph_generation.greenhouse import Greenhouse  # noqa: E402

OUT = os.path.join(HERE, "..", "tests", "golden", "sim_golden.npz")
CONFIG = "/root/reference/docker/vessel_graph_gen_docker_config.yml"


def run_reference(cfg, seed):
    random.seed(seed)
    np.random.seed(seed)
    gh = Greenhouse(cfg["Greenhouse"])
    af = Forest(cfg["Forest"], gh.d, gh.r, gh.simspace, nerve_center=gh.nerve_center, nerve_radius=gh.nerve_radius)
    vf = Forest(cfg["Forest"], gh.d, gh.r, gh.simspace, arterial=False, nerve_center=gh.nerve_center,
                nerve_radius=gh.nerve_radius)
    gh.set_forests(af, vf)
    gh.develop_forest()
    buf = io.StringIO(newline="")
    w = csv.writer(buf)
    w.writerow(["node1", "node2", "radius"])
    n_art = 0
    for k, forest in enumerate((af, vf)):
        for tree in forest.get_trees():
            for n in tree.get_tree_iterator(exclude_root=True, only_active=False):
                w.writerow([n.position, n.get_proximal_node().position, n.radius])
                n_art += (k == 0)
    trace = np.array([gh.art_nodes_per_step[1:], gh.oxys_per_step[1:], gh.ven_nodes_per_step[1:],
                      gh.co2_per_step[1:]], dtype=np.int64).T
    oxy = np.array(gh.oxy_mesh.get_all_elements(), dtype=np.float64).reshape(-1, 3)
    co2 = np.array(gh.co2_mesh.get_all_elements(), dtype=np.float64).reshape(-1, 3)
    return dict(csv=buf.getvalue(), trace=trace, oxy=oxy, co2=co2, faz=float(gh.FAZ_radius), n_art=n_art,
                next_py=random.random(), next_np=float(np.random.random_sample()))


def main():
    base = yaml.safe_load(open(CONFIG))
    g = {"config_yaml": np.array(yaml.safe_dump(base))}
    cases = [(0, 30, 20), (1, 30, 20), (2, 30, 20), (3, 30, 20), (0, 10, 5), (5, 10, 5), (11, 20, 0), (4, 0, 12)]
    if "--full" in sys.argv:
        cases += [(0, 100, 150), (7, 100, 150)]
    keep_old = {}
    if os.path.exists(OUT) and "--full" not in sys.argv:
        old = np.load(OUT)
        keep_old = {k: old[k] for k in old.files if k.startswith("full")}
    names = []
    for seed, i1, i2 in cases:
        cfg = copy.deepcopy(base)
        cfg["Greenhouse"]["modes"][0]["I"] = i1
        cfg["Greenhouse"]["modes"][1]["I"] = i2
        r = run_reference(cfg, seed)
        full = i1 + i2 >= 250
        name = f"{'full' if full else 'run'}_s{seed}_{i1}_{i2}"
        names.append(name)
        g[name + "_seed_I"] = np.array([seed, i1, i2])
        g[name + "_trace"] = r["trace"]
        g[name + "_faz"] = np.array(r["faz"])
        g[name + "_n_art"] = np.array(r["n_art"])
        g[name + "_next"] = np.array([r["next_py"], r["next_np"]])
        g[name + "_csv_sha256"] = np.array(hashlib.sha256(r["csv"].encode()).hexdigest())
        if not full:
            g[name + "_csv"] = np.frombuffer(r["csv"].encode(), dtype=np.uint8)
            g[name + "_oxy"] = r["oxy"]
            g[name + "_co2"] = r["co2"]
        print(name, "rows", r["csv"].count("\n") - 1, "oxy", len(r["oxy"]), "co2", len(r["co2"]))
    # f4: optic-nerve forests (forest.py:38-66) with the nerve disc inside the field of view (simulation_space.py:48-50): the
    # notebook's 12x12 mm^2 geometry (example_custom_vessel_simulation.ipynb:138-156: param_scale 12, 16 trees, N = 8000, thinner z),
    # short runs
    nerve = copy.deepcopy(base)
    nerve["Greenhouse"]["param_scale"] = 12
    nerve["Forest"]["type"] = "nerve"
    nerve["Forest"]["N_trees"] = 16
    nerve["Greenhouse"]["SimulationSpace"]["no_voxel_z"] = 0.0033
    nerve["Greenhouse"]["d"] = 0.15
    for m in nerve["Greenhouse"]["modes"]:
        m["N"] = 8000
        m["delta_sigma"] = 0.002222
    g["nerve_config_yaml"] = np.array(yaml.safe_dump(nerve))
    for seed, i1, i2 in [(0, 12, 6), (3, 12, 6), (8, 20, 0)]:
        cfg = copy.deepcopy(nerve)
        cfg["Greenhouse"]["modes"][0]["I"] = i1
        cfg["Greenhouse"]["modes"][1]["I"] = i2
        r = run_reference(cfg, seed)
        name = f"nerve_s{seed}_{i1}_{i2}"
        names.append(name)
        g[name + "_seed_I"] = np.array([seed, i1, i2])
        g[name + "_trace"] = r["trace"]
        g[name + "_faz"] = np.array(r["faz"])
        g[name + "_n_art"] = np.array(r["n_art"])
        g[name + "_next"] = np.array([r["next_py"], r["next_np"]])
        g[name + "_csv"] = np.frombuffer(r["csv"].encode(), dtype=np.uint8)
        g[name + "_oxy"] = r["oxy"]
        g[name + "_co2"] = r["co2"]
        print(name, "rows", r["csv"].count("\n") - 1, "oxy", len(r["oxy"]), "co2", len(r["co2"]))
    # f4: fixed sampling geometry (simulation_space.py:29-34, 70-76) with the mask the reference ships
    # (vessel_graph_generation/geometries/slab_oxy_sample_3mm.npy, [76, 76, 1]); the mask itself is stored as data
    geo_path = "/root/reference/vessel_graph_generation/geometries/slab_oxy_sample_3mm.npy"
    g["geometry_mask"] = np.load(geo_path)
    for seed, i1, i2 in [(0, 30, 20), (6, 10, 5)]:
        cfg = copy.deepcopy(base)
        cfg["Greenhouse"]["SimulationSpace"]["oxygen_sample_geometry_path"] = geo_path
        cfg["Greenhouse"]["modes"][0]["I"] = i1
        cfg["Greenhouse"]["modes"][1]["I"] = i2
        r = run_reference(cfg, seed)
        name = f"geom_s{seed}_{i1}_{i2}"
        names.append(name)
        g[name + "_seed_I"] = np.array([seed, i1, i2])
        g[name + "_trace"] = r["trace"]
        g[name + "_faz"] = np.array(r["faz"])
        g[name + "_n_art"] = np.array(r["n_art"])
        g[name + "_csv"] = np.frombuffer(r["csv"].encode(), dtype=np.uint8)
        g[name + "_oxy"] = r["oxy"]
        g[name + "_co2"] = r["co2"]
        print(name, "rows", r["csv"].count("\n") - 1, "oxy", len(r["oxy"]), "co2", len(r["co2"]))
    for k, v in keep_old.items():
        g.setdefault(k, v)
        if k.endswith("_seed_I"):
            names.append(k[: -len("_seed_I")])
    g["names"] = np.array(sorted(set(names)))
    np.savez_compressed(OUT, **g)
    print("wrote", os.path.abspath(OUT), os.path.getsize(OUT) // 1024, "KiB")


if __name__ == "__main__":
    main()
