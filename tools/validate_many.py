"""One-off validation: N full-length samples on the GPU vs the CPU oracle (CSV text + radii bits).
  python tools/validate_many.py [N=64] [first_seed=1000]
  python tools/validate_many.py --digests tools/cache/oracle_digests_1000_512.npz [--reps R]
The second form compares against per-seed SHA-256 values the oracle produced in the build container (tools/oracle_digests.py):
every double of every edge list and the CSV text, without spending GPU-box time on the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, yaml
from multiprocessing import Pool


def oracle_one(args):
    cfg, seed = args
    from oracle import sim_oracle
    e, info = sim_oracle.simulate(cfg, seed)
    return seed, e, info["n_art_edges"]


def against_digests(path, reps):
    import hashlib
    from octa_autosegmentation_amd.utils import configs
    from octa_autosegmentation_amd import graph_io
    from octa_autosegmentation_amd.vessel_graph_generation import greenhouse
    d = np.load(path)
    seeds = [int(v) for v in d["seeds"]]
    cfg = configs.load_generator_config()
    sim = greenhouse.BatchSimulator(cfg, len(seeds))
    failed = 0
    for rep in range(reps):
        t = time.time()
        res = sim.run(seeds)
        el = time.time() - t
        bad_d = bad_t = 0
        err = int(np.asarray(res.stats)[:, 0].max())
        if err:
            print(f"rep {rep}: the simulator reported error / capacity bits 0x{err:x}")
            failed += 1
        for k, seed in enumerate(seeds):
            e = np.ascontiguousarray(res.sample_edges(k))
            if e.shape[0] != int(d["rows"][k]) or hashlib.sha256(e.tobytes()).hexdigest() != str(d["sha_doubles"][k]):
                bad_d += 1
                if hashlib.sha256(graph_io.edges_to_csv_text(e).encode()).hexdigest() != str(d["sha_text"][k]):
                    bad_t += 1
                    print("seed", seed, "CSV text differs from the oracle's")
        print(f"RESULT rep {rep}: {len(seeds)} full-length samples ({seeds[0]}..{seeds[-1]}) in {el:.2f} s: samples with differing doubles {bad_d}, "
              f"with differing CSV text {bad_t}", flush=True)
        failed += int(bad_d > 0 or bad_t > 0)
    sim.close()
    return failed


if __name__ == "__main__":
    if "--digests" in sys.argv:
        a = sys.argv
        bad_reps = against_digests(a[a.index("--digests") + 1], int(a[a.index("--reps") + 1]) if "--reps" in a else 1)
        sys.exit(1 if bad_reps else 0)          # a mismatch (or an error bit) in any repetition fails the command
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    from octa_autosegmentation_amd.utils import configs
    cfg = configs.load_generator_config()
    seeds = list(range(s0, s0 + N))
    t = time.time()
    with Pool(min(N, os.cpu_count() or 1, 64)) as p:
        ref = p.map(oracle_one, [(cfg, s) for s in seeds])
    print(f"oracle: {N} samples in {time.time()-t:.1f} s")
    import torch
    from octa_autosegmentation_amd import graph_io
    from octa_autosegmentation_amd.vessel_graph_generation import greenhouse
    t = time.time()
    res = greenhouse.simulate_batch(cfg, seeds)
    print(f"gpu: {N} samples in {time.time()-t:.1f} s; error bits {int(res.stats[:,0].max())}")
    bad_text = bad_rad = bad_bits = 0
    n_diff = n_vals = 0
    for k, (seed, e, na) in enumerate(ref):
        gpu = res.sample_edges(k)
        ok_shape = gpu.shape == e.shape and res.n_art[k] == na
        if not ok_shape or graph_io.edges_to_csv_text(gpu) != graph_io.edges_to_csv_text(e):
            bad_text += 1
            print("seed", seed, "CSV text differs", gpu.shape, e.shape)
            continue
        bad_rad += int(not (gpu[:, 6] == e[:, 6]).all())
        bad_bits += int(not (gpu == e).all())
        n_diff += int((gpu != e).sum()); n_vals += gpu.size
        if not (gpu == e).all():
            rows = np.nonzero((gpu != e).any(axis=1))[0]
            print("seed", seed, "differing doubles", int((gpu != e).sum()), "first rows", rows[:4], "first diff", (gpu[rows[0]] - e[rows[0]]))
    print(f"RESULT: {N} full-length samples: CSV text mismatches {bad_text}, radius-bit mismatches {bad_rad}, "
          f"samples whose position doubles differ in the last bits (same text) {bad_bits}; differing doubles {n_diff} of {n_vals}")
