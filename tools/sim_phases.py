"""Per-phase device time of the simulator: one 128-sample full-length launch with the GPU to itself; 100 MHz timers of thread 0
(octa_sim_stats slots 8..23 = prof[0..15], 24..31 = kdprof[0..7]). usage: python tools/sim_phases.py [batch] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from octa_autosegmentation_amd.utils import configs  # noqa: E402
from octa_autosegmentation_amd.vessel_graph_generation import greenhouse  # noqa: E402

NAMES = ["sample", "assign_art", "pre_art", "seq_art", "satisfy_art", "mailbox", "assign_ven", "pre_ven", "seq_ven", "satisfy_ven",
         "candidates*", "kd_total*", "pairs+ven*", "pair_sort*", "set_replay*", "compact*"]
KD = ["bbox", "dim", "gather", "nth_wave", "nth_quarter", "next_level", "finalize", "murray(seq)*"]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    cfg = configs.load_generator_config()
    sim = greenhouse.BatchSimulator(cfg, B)
    for rep in range(reps):
        t0 = time.time()
        res = sim.run(np.arange(B) + 5000 + 1000 * rep)
        dt = time.time() - t0
        st = res.stats.astype(np.float64)
        assert int(st[:, 0].max()) == 0
        ph = st[:, 8:24].mean(axis=0) * 1e-5
        kd = st[:, 24:32].mean(axis=0) * 1e-5
        print(f"rep {rep}: wall {dt * 1e3:.0f} ms, kernel {res.timing['kernel_b_ms']:.0f} ms, per-sample phases sum {ph[:10].sum():.1f} ms")
        print("  " + "  ".join(f"{n} {v:.1f}" for n, v in zip(NAMES, ph)))
        print("  kd: " + "  ".join(f"{n} {v:.1f}" for n, v in zip(KD, kd)))
        if os.environ.get("OCTA_PHASES_RAW"):     # diagnostic builds keep counts in the kd slots: raw means, and the samples' scalar statistics
            print("  kd raw: " + "  ".join(f"{v:.0f}" for v in st[:, 24:32].mean(axis=0)))
            print("  stats[0:8] (err, py_pos, murray_steps, n_bif, respec, ...): " + "  ".join(f"{v:.0f}" for v in st[:, 0:8].mean(axis=0)))
    sim.close()


if __name__ == "__main__":
    main()
