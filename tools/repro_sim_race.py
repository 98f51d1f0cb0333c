"""GPU only: repeat a 512-sample batch; for every sample whose per-iteration trace differs from the first run's, print the first
iteration that differs and the counts around it.

    python tools/repro_sim_race.py [reps] [--seed0 N] [--batch B]

With a diagnostic build of the library (tools/build_sim_variant.py NAME -DOCTA_SIM_DEBUG_SAT ..., selected with OCTA_HIP_LIB) the
kernel also writes a 16-word digest of every phase_satisfy_art call; the digest row of the first differing iteration is printed
next to the first run's, which names the stage that went wrong (columns below)."""
import os
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
DIGEST = ["n_new", "n_oxy", "n_pairs", "removed", "removed&ven", "pair_sinks", "pair_sinks_taken", "pairs!=removed", "n_ins", "mask", "appended",
          "slots", "in_distinct", "bad_slots", "n_co2_in", "n_co2_out"]


def main():
    args = sys.argv[1:]
    nrep = int(args[0]) if args and not args[0].startswith("--") else 30
    seed0 = int(args[args.index("--seed0") + 1]) if "--seed0" in args else 90000
    batch = int(args[args.index("--batch") + 1]) if "--batch" in args else 512
    dump = os.path.join(tempfile.gettempdir(), f"octa_sim_dbg_{os.getpid()}.bin")
    os.environ["OCTA_SIM_DEBUG_DUMP"] = dump
    from octa_autosegmentation_amd.utils import configs
    from octa_autosegmentation_amd.vessel_graph_generation import greenhouse
    cfg = configs.load_generator_config()
    sim = greenhouse.BatchSimulator(cfg, batch)
    seeds = np.arange(batch) + seed0
    ref = ref_dbg = None
    events = relaunches = parked = 0
    t0 = time.time()
    kernel_ms = []
    for rep in range(nrep):
        if os.path.exists(dump):
            os.remove(dump)
        res = sim.run(seeds)
        tr = sim.trace().copy()
        kernel_ms.append(res.timing["kernel_b_ms"])
        relaunches += res.service["relaunches"]
        parked += res.service["parked"]
        dbg = None
        if os.path.exists(dump):
            dbg = np.fromfile(dump, np.int32).reshape(batch, -1, 16)
        if ref is None:
            ref, ref_dbg = tr, dbg
            continue
        bad = np.flatnonzero((tr != ref).any(axis=(1, 2)))
        for k in bad:
            it = int(np.flatnonzero((tr[k] != ref[k]).any(axis=1))[0])
            events += 1
            print(f"rep {rep} sample {k}: first differing iteration {it}: ref {ref[k][max(it - 1, 0):it + 2].tolist()} got {tr[k][max(it - 1, 0):it + 2].tolist()}"
                  f" (this run: relaunches {res.service['relaunches']}, parked {res.service['parked']})", flush=True)
            if dbg is not None:
                dit = int(np.flatnonzero((dbg[k] != ref_dbg[k]).any(axis=1))[0]) if (dbg[k] != ref_dbg[k]).any() else -1
                print(f"    first differing digest row: iteration {dit}")
                if dit >= 0:
                    for name, a, c in zip(DIGEST, ref_dbg[k][dit], dbg[k][dit]):
                        print(f"      {name:18s} ref {a:8d} got {c:8d}{'   <--' if a != c else ''}")
        if dbg is not None:
            # in-kernel consistency of every call of this run, event or not: appended == sinks taken by the pairs, table slots == appended ...
            d = dbg.reshape(-1, 16)
            d = d[d[:, 0] >= 0]
            incons = np.flatnonzero((d[:, 10] != d[:, 6]) | (d[:, 7] != 0) | (d[:, 13] > 0) | ((d[:, 11] >= 0) & (d[:, 11] != d[:, 10])) | ((d[:, 12] >= 0) & (d[:, 12] != d[:, 6]))
                                    | (d[:, 15] - d[:, 14] != d[:, 10]) | (d[:, 5] != d[:, 3]))
            for r in incons[:8]:
                print(f"rep {rep}: INCONSISTENT digest row: " + ", ".join(f"{n}={v}" for n, v in zip(DIGEST, d[r])), flush=True)
    el = time.time() - t0
    print(f"events {events} in {(nrep - 1) * batch} sample runs; relaunches {relaunches}, parked workgroups {parked}; "
          f"kernel {np.mean(kernel_ms):.1f} ms per launch (min {np.min(kernel_ms):.1f}); {el:.0f} s wall")
    sim.close()


if __name__ == "__main__":
    main()
