"""The reference's notebook configuration (12 x 12 mm^2, I = 400 + 500, N = 8000, optic-nerve forests with 16 trees) on the GPU's
wide-field build, full length: rows, SHA-256 of the CSV text, peaks of the per-iteration statistics, device time.
  python tools/run_f4_full.py [seed ...]      (default 0; all seeds run as ONE batch)
Compares with tests/golden/sim_f4_golden.npz where a seed has a reference-made entry."""
import hashlib
import os
import sys
import time

import numpy as np
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from octa_autosegmentation_amd import graph_io  # noqa: E402
from octa_autosegmentation_amd.vessel_graph_generation import greenhouse  # noqa: E402


def notebook_config():
    g = np.load(os.path.join(ROOT, "tests", "golden", "sim_golden.npz"))
    cfg = yaml.safe_load(str(g["nerve_config_yaml"]))
    cfg["Greenhouse"]["modes"][0]["I"] = int(os.environ.get("F4_I1", "400"))
    cfg["Greenhouse"]["modes"][1]["I"] = int(os.environ.get("F4_I2", "500"))
    return cfg


def main():
    seeds = [int(a) for a in sys.argv[1:]] or [0]
    cfg = notebook_config()
    gold = None
    gp = os.path.join(ROOT, "tests", "golden", "sim_f4_golden.npz")
    if os.path.exists(gp):
        gold = np.load(gp)
    sim = greenhouse.BatchSimulator(cfg, len(seeds))
    print("wide-field build:", sim.is_large, flush=True)
    t0 = time.time()
    try:
        res = sim.run(seeds)
    except Exception as e:      # a capacity error: show how far every sample came
        print("run failed:", e, flush=True)
        tr = sim.trace()
        for k, seed in enumerate(seeds):
            nz = np.flatnonzero(tr[k][:, 0])
            last = int(nz[-1]) if len(nz) else -1
            print(f"seed {seed}: last recorded iteration {last}: {tr[k][max(last - 2, 0):last + 1].tolist()}", flush=True)
        sim.close()
        return
    dt = time.time() - t0
    tr = sim.trace()
    print(f"{len(seeds)} samples in {dt:.1f} s (kernel {res.timing['kernel_b_ms'] / 1e3:.1f} s, relaunches {res.service['relaunches']}, "
          f"bifurcation requests {res.timing['bif_requests']})", flush=True)
    for k, seed in enumerate(seeds):
        e = res.sample_edges(k)
        text = graph_io.edges_to_csv_text(e)
        sha = hashlib.sha256(text.encode()).hexdigest()
        print(f"seed {seed}: error bits {int(res.stats[k, 0])}, rows {len(e)}, peaks (art, O2, ven, CO2) {tr[k].max(axis=0).tolist()}, sha256 {sha}", flush=True)
        if gold is not None and f"s{seed}_csv_sha256" in gold.files:
            ok_t = bool((tr[k] == gold[f"s{seed}_trace"]).all()) if tr[k].shape == gold[f"s{seed}_trace"].shape else False
            first = None
            if not ok_t and tr[k].shape == gold[f"s{seed}_trace"].shape:
                first = int(np.argwhere((tr[k] != gold[f"s{seed}_trace"]).any(axis=1))[0, 0])
            print(f"   reference: rows {int(gold[f's{seed}_rows'])}, sha256 equal {sha == str(gold[f's{seed}_csv_sha256'])}, trace equal {ok_t}"
                  + (f" (first differing iteration {first}: gpu {tr[k][first].tolist()} ref {gold[f's{seed}_trace'][first].tolist()})" if first is not None else ""), flush=True)
    sim.close()


if __name__ == "__main__":
    main()
