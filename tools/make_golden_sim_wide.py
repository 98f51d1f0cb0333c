"""tools/make_golden_sim_wide.py -- generates tests/golden/sim_wide_golden.npz (round 3).

Widens the reference pin of the simulator from "two full-length SHA fixtures" to N full-length
seeds. Runs ONLY in the build container: every seed is run through the imported reference
(/root/reference, `anytree` stand-in as in make_golden_sim.py; `random.seed(s); np.random.seed(s)`)
with the shipped docker config at full length (I = 100 + 150), and through the oracle; stored per seed:

  seeds            int64[N]
  rows             int64[N]      CSV data rows of the reference run
  csv_sha256       str[N]        SHA-256 of the reference's CSV text
  trace_sha256     str[N]        SHA-256 of the (art nodes, O2, ven nodes, CO2)-per-iteration int64 trace
  oracle_text_equal  bool[N]     oracle's CSV text == reference's CSV text
  oracle_diff_doubles int64[N]   doubles of the oracle's edge list that differ from the reference's (positions + radii;
                                 -1 when the row counts differ)
  ref_seconds      float64[N]    wall time of the reference run (one core)

The reference here runs on an AVX-512 host, where numpy's `np.arccos` is its own SIMD kernel; the oracle
follows glibc's `acos`, so `oracle_diff_doubles` is expected to be small and non-zero for
some seeds while `oracle_text_equal` says whether that ever reaches a printed digit.

  python tools/make_golden_sim_wide.py [--seeds 64] [--first 1000] [--workers 4]
"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden", "sim_wide_golden.npz")
CONFIG = "/root/reference/docker/vessel_graph_gen_docker_config.yml"


def run_reference_once(cfg, seed, mg):
    """One seeded run of the reference's objects: CSV text exactly as its export block writes it
    (generate_vessel_graph.py:43-66), the per-iteration trace, and the edge list as doubles ([n,7], CSV row order)."""
    import csv
    import io
    import random
    random.seed(seed)
    np.random.seed(seed)
    gh = mg.Greenhouse(cfg["Greenhouse"])
    af = mg.Forest(cfg["Forest"], gh.d, gh.r, gh.simspace, nerve_center=gh.nerve_center, nerve_radius=gh.nerve_radius)
    vf = mg.Forest(cfg["Forest"], gh.d, gh.r, gh.simspace, arterial=False, nerve_center=gh.nerve_center,
                   nerve_radius=gh.nerve_radius)
    gh.set_forests(af, vf)
    gh.develop_forest()
    buf = io.StringIO(newline="")
    w = csv.writer(buf)
    w.writerow(["node1", "node2", "radius"])
    rows = []
    for forest in (af, vf):
        for tree in forest.get_trees():
            for n in tree.get_tree_iterator(exclude_root=True, only_active=False):
                w.writerow([n.position, n.get_proximal_node().position, n.radius])
                rows.append(list(n.position) + list(n.get_proximal_node().position) + [n.radius])
    trace = np.array([gh.art_nodes_per_step[1:], gh.oxys_per_step[1:], gh.ven_nodes_per_step[1:],
                      gh.co2_per_step[1:]], dtype=np.int64).T
    return buf.getvalue(), trace, np.array(rows, dtype=np.float64).reshape(-1, 7)


def one(seed):
    sys.path.insert(0, HERE)
    sys.path.insert(0, ROOT)
    import make_golden_sim as mg          # imports the reference
    from oracle import sim_oracle
    cfg = yaml.safe_load(open(CONFIG))
    t0 = time.time()
    text, trace, ref_edges = run_reference_once(cfg, seed, mg)
    dt = time.time() - t0
    edges, info = sim_oracle.simulate(cfg, seed)
    otext = sim_oracle.edges_to_csv_text(edges)
    ndiff = int((ref_edges != edges).sum()) if ref_edges.shape == edges.shape else -1
    return dict(seed=seed, rows=text.count("\n") - 1, sha=hashlib.sha256(text.encode()).hexdigest(),
                tsha=hashlib.sha256(np.ascontiguousarray(trace).tobytes()).hexdigest(),
                equal=(otext == text), ndiff=ndiff, secs=dt,
                trace_equal=bool(info["trace"].shape == trace.shape and (info["trace"] == trace).all()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=64)
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--workers", type=int, default=4)
    a = ap.parse_args()
    seeds = list(range(a.first, a.first + a.seeds))
    from concurrent.futures import ProcessPoolExecutor
    res = []
    with ProcessPoolExecutor(max_workers=a.workers) as ex:
        for r in ex.map(one, seeds):
            print(r, flush=True)
            res.append(r)
    g = dict(seeds=np.array([r["seed"] for r in res], dtype=np.int64),
             rows=np.array([r["rows"] for r in res], dtype=np.int64),
             csv_sha256=np.array([r["sha"] for r in res]),
             trace_sha256=np.array([r["tsha"] for r in res]),
             oracle_text_equal=np.array([r["equal"] for r in res]),
             oracle_trace_equal=np.array([r["trace_equal"] for r in res]),
             oracle_diff_doubles=np.array([r["ndiff"] for r in res], dtype=np.int64),
             ref_seconds=np.array([r["secs"] for r in res]))
    np.savez_compressed(OUT, **g)
    print("wrote", OUT, "text-equal", int(g["oracle_text_equal"].sum()), "of", len(seeds),
          "differing doubles per seed", g["oracle_diff_doubles"].tolist())


if __name__ == "__main__":
    main()
