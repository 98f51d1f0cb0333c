"""GPU only, diagnostic build: per-ITERATION device time of the simulator's phases -- which iterations are expensive.

    python tools/build_sim_variant.py iterprof -DOCTA_SIM_DEBUG_SAT -DOCTA_SIM_ITER_PROF
    OCTA_HIP_LIB=gpurun_variants/liboctahip_iterprof.so python tools/sim_iter_profile.py [batch]

One full-length launch; for every phase timer (tools/sim_phases.py's names) the mean over the samples of the time spent per iteration,
printed as: total, the share of the ten most expensive iterations, and those iterations."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NAMES = ["sample", "assign_art", "pre_art", "seq_art", "satisfy_art", "mailbox", "assign_ven", "pre_ven", "seq_ven", "satisfy_ven",
         "candidates*", "kd_total*", "pairs+ven*", "pair_sort*", "set_replay*", "compact*"]


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    dump = os.path.join(tempfile.gettempdir(), f"octa_sim_dbg_{os.getpid()}.bin")
    os.environ["OCTA_SIM_DEBUG_DUMP"] = dump
    from octa_autosegmentation_amd.utils import configs
    from octa_autosegmentation_amd.vessel_graph_generation import greenhouse
    sim = greenhouse.BatchSimulator(configs.load_generator_config(), batch)
    res = sim.run(np.arange(batch) + 5000)
    assert int(res.stats[:, 0].max()) == 0
    d = np.fromfile(dump, np.int32).reshape(batch, -1, 16).astype(np.float64) * 1e-5      # ms
    d = np.where(d < 0, 0.0, d)
    m = d.mean(axis=0)                 # [iterations + 1][16]
    print(f"{batch} samples, {m.shape[0]} rows; per-sample total {m[:, :10].sum():.1f} ms")
    for k, name in enumerate(NAMES):
        col = m[:, k]
        top = np.argsort(-col)[:10]
        print(f"{name:13s} total {col.sum():7.2f} ms   ten most expensive iterations {col[top].sum():6.2f} ms ({100 * col[top].sum() / max(col.sum(), 1e-9):4.1f} %): "
              + ", ".join(f"{i}:{col[i]:.2f}" for i in top[:6]))
    tot = m[:, :10].sum(axis=1)
    top = np.argsort(-tot)[:12]
    print("iterations by total: " + ", ".join(f"{i}:{tot[i]:.2f}" for i in top) + f"   median {np.median(tot):.2f} ms")
    sim.close()


if __name__ == "__main__":
    main()
