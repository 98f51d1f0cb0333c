"""Build container (CPU): run the oracle on a range of full-length seeds and store, per seed, SHA-256 of the edge list's doubles and of
its CSV text. tools/validate_many.py --digests FILE then checks GPU batches against them without spending GPU-box time on the oracle.
  python tools/oracle_digests.py FIRST_SEED N OUT.npz [workers]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from multiprocessing import Pool


def one(args):
    cfg, seed = args
    from oracle import sim_oracle
    e, info = sim_oracle.simulate(cfg, seed)
    e = np.ascontiguousarray(e, dtype=np.float64)
    text = sim_oracle.edges_to_csv_text(e)
    return seed, e.shape[0], int(info["n_art_edges"]), hashlib.sha256(e.tobytes()).hexdigest(), hashlib.sha256(text.encode()).hexdigest()


if __name__ == "__main__":
    s0, n, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
    workers = int(sys.argv[4]) if len(sys.argv) > 4 else (os.cpu_count() or 1)
    from octa_autosegmentation_amd.utils import configs
    cfg = configs.load_generator_config()
    t = time.time()
    with Pool(workers) as p:
        rows = p.map(one, [(cfg, s) for s in range(s0, s0 + n)], chunksize=1)
    np.savez_compressed(out, seeds=np.array([r[0] for r in rows]), rows=np.array([r[1] for r in rows]), n_art=np.array([r[2] for r in rows]),
                        sha_doubles=np.array([r[3] for r in rows]), sha_text=np.array([r[4] for r in rows]))
    print(f"{n} seeds in {time.time() - t:.0f} s -> {out}")
