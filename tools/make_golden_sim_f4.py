"""tools/make_golden_sim_f4.py -- generates tests/golden/sim_f4_golden.npz: the reference's notebook configuration
(example_custom_vessel_simulation.ipynb:138-156: 12 x 12 mm^2 field of view -- param_scale 12, optic-nerve forests with 16 trees,
N = 8000 candidates per iteration, I = 400 + 500 iterations, no_voxel_z 0.0033, d 0.15, delta_sigma 0.002222) run at FULL length
through the imported reference (tools/make_golden_sim_wide.run_reference_once; build container only, about nine minutes per seed).
Stored per seed: rows, SHA-256 of the CSV text, the per-iteration trace (arterial nodes, O2 sinks, venous nodes, CO2 sources),
the peak counts (what a device build must hold), the FAZ radius and the wall time.

  python tools/make_golden_sim_f4.py [seed ...]        (default: 0)
"""
import hashlib
import os
import sys
import time

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "sim_f4_golden.npz")


def notebook_config():
    g = np.load(os.path.join(ROOT, "tests", "golden", "sim_golden.npz"))
    cfg = yaml.safe_load(str(g["nerve_config_yaml"]))          # the notebook's overrides on the shipped docker config (make_golden_sim.py)
    cfg["Greenhouse"]["modes"][0]["I"] = 400
    cfg["Greenhouse"]["modes"][1]["I"] = 500
    return cfg


def one(seed):
    import make_golden_sim as mg
    from make_golden_sim_wide import run_reference_once
    cfg = notebook_config()
    t0 = time.time()
    text, trace, edges = run_reference_once(cfg, seed, mg)
    dt = time.time() - t0
    print(f"seed {seed}: {text.count(chr(10)) - 1} rows, peaks (art, O2, ven, CO2) {trace.max(axis=0).tolist()}, {dt:.0f} s", flush=True)
    return seed, {"rows": np.array(text.count("\n") - 1), "csv_sha256": np.array(hashlib.sha256(text.encode()).hexdigest()), "trace": trace,
                  "peaks": trace.max(axis=0), "seconds": np.array(dt),
                  "edges_sha256": np.array(hashlib.sha256(np.ascontiguousarray(edges).tobytes()).hexdigest())}


def main():
    """Seeds given on the command line are ADDED to the existing file (one worker process per seed, at most three at a time)."""
    from concurrent.futures import ProcessPoolExecutor
    seeds = [int(a) for a in sys.argv[1:]] or [0]
    out = {}
    if os.path.exists(OUT):
        old = np.load(OUT)
        out = {k: old[k] for k in old.files}
    out["config_yaml"] = np.array(yaml.safe_dump(notebook_config()))
    with ProcessPoolExecutor(max_workers=min(3, len(seeds))) as ex:
        for seed, d in ex.map(one, seeds):
            for k, v in d.items():
                out[f"s{seed}_{k}"] = v
    have = sorted({int(k[1:].split("_")[0]) for k in out if k.startswith("s") and k[1:2].isdigit() and k.endswith("_rows")})
    out["seeds"] = np.array(have, dtype=np.int64)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "seeds", have)


if __name__ == "__main__":
    main()
