"""Experiment (round 6): the headline pipeline with the GPU PARTITIONED by CU masks instead of ordered by host-side polling.
Simulator launches go to streams whose queue may only use the `sim` CUs, rasterisations to a stream on the remaining CUs; several
launches in flight, no gate, no sleeps. Usage: python tools/exp_cumask_pipeline.py --raster-cus 32 --policy stride --inflight 3 --steps 24"""
import argparse, ctypes, os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_masks(n_cus, n_raster, policy):
    bits = np.zeros(n_cus, bool)
    if n_raster > 0:
        if policy == "stride":
            idx = (np.arange(n_raster) * n_cus // n_raster)
        elif policy == "tail":
            idx = np.arange(n_cus - n_raster, n_cus)
        elif policy == "head":
            idx = np.arange(n_raster)
        bits[idx] = True
    def words(b):
        w = np.zeros((n_cus + 31) // 32, np.uint32)
        for i in np.nonzero(b)[0]:
            w[i // 32] |= np.uint32(1 << (i % 32))
        return w
    return words(~bits), words(bits)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--raster-cus", type=int, default=32)
    ap.add_argument("--policy", default="stride")
    ap.add_argument("--inflight", type=int, default=3)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--group", type=int, default=4)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--sim-grid", type=int, default=0)
    a = ap.parse_args()
    import torch
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n_cus = torch.cuda.get_device_properties(0).multi_processor_count
    sim_mask, ras_mask = make_masks(n_cus, a.raster_cus, a.policy)
    os.environ["OCTA_SIM_GRID"] = str(a.sim_grid or 2 * (n_cus - a.raster_cus))
    hip = ctypes.CDLL("libamdhip64.so")
    def mk(mask):
        st = ctypes.c_void_p()
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(len(mask)), mask.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0, rc
        return torch.cuda.ExternalStream(st.value, device=dev)
    from octa_autosegmentation_amd import pipeline
    from octa_autosegmentation_amd.utils import sharding
    import yaml
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "docker", "vessel_graph_gen_docker_config.yml")))
    B, G, nf = a.batch, a.group, a.inflight
    gens = [pipeline.TripleGenerator(cfg, B * G) for _ in range(nf)]
    sim_streams = [mk(sim_mask) if a.raster_cus > 0 else torch.cuda.Stream() for _ in range(nf)]
    ras_streams = [mk(ras_mask) if a.raster_cus > 0 else torch.cuda.Stream() for _ in range(nf)]
    spans = []
    def launch(slot, first, n):
        seeds = np.concatenate([sharding.rank_seeds(0, i, B) for i in range(first, first + n)])
        torch.cuda.set_device(dev)
        g = gens[slot]
        t0 = time.time()
        with torch.cuda.stream(sim_streams[slot]):
            res = g.sim.run(seeds)
        t1 = time.time()
        with torch.cuda.stream(ras_streams[slot]):
            out = g._render(res, True)
            t2 = time.time()
            ras_streams[slot].synchronize()
        t3 = time.time()
        spans.append((slot, t0, t1, t2, t3, res.timing["kernel_b_ms"]))
        return out
    def run_steps(first, count):
        groups = [(first + k, min(G, count - k)) for k in range(0, count, G)]
        chains = [[g for j, g in enumerate(groups) if j % nf == s] for s in range(nf)]
        ths = [threading.Thread(target=lambda ch=ch, s=s: [launch(s, f0, n) for f0, n in ch]) for s, ch in enumerate(chains)]
        for t in ths: t.start()
        for t in ths: t.join()
    run_steps(0, a.warmup)
    torch.cuda.synchronize()
    spans.clear()
    t0 = time.time()
    run_steps(a.warmup, a.steps)
    torch.cuda.synchronize()
    dt = time.time() - t0
    k = np.array([s[5] for s in spans])
    print(f"raster_cus={a.raster_cus} policy={a.policy} inflight={nf} grid={os.environ['OCTA_SIM_GRID']}: {B * a.steps / dt:.1f} samples/s; "
          f"kernel ms mean {k.mean():.1f} min {k.min():.1f} max {k.max():.1f}; sim call {1e3 * np.mean([s[2] - s[1] for s in spans]):.1f} ms, "
          f"render enqueue {1e3 * np.mean([s[3] - s[2] for s in spans]):.1f} ms, render wait {1e3 * np.mean([s[4] - s[3] for s in spans]):.1f} ms", flush=True)


if __name__ == "__main__":
    main()
