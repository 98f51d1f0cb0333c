#!/usr/bin/env python3
"""Host-side timeline of the headline leg's launches (bench.py's slot-affine chains): per launch, when its slot asked for the
simulator gate, got it, left the simulator call, finished enqueueing the rasterisation, and saw its stream drain.

    python tools/slot_timeline.py [--inflight 3] [--steps 48] [--warmup 12]
"""
import argparse, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inflight", type=int, default=2)
    ap.add_argument("--steps", type=int, default=48)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--group", type=int, default=4)
    args = ap.parse_args()
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from octa_autosegmentation_amd import pipeline
    from octa_autosegmentation_amd.utils import configs, sharding
    cfg = configs.load_generator_config()
    dev = torch.device("cuda", 0)
    B, G, n_fly = 128, args.group, args.inflight
    gens = [pipeline.TripleGenerator(cfg, B * G) for _ in range(n_fly)]
    streams = [torch.cuda.Stream() for _ in range(n_fly)]
    gate = pipeline.SimGate()
    for g in gens:
        g.sim_gate = gate

    def launch(slot, first):
        seeds = np.concatenate([sharding.rank_seeds(0, i, B) for i in range(first, first + G)])
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[slot]):
            out = gens[slot].generate(seeds)
            streams[slot].synchronize()
        w = out["wall"]
        w["t_done"] = time.time(); w["slot"] = slot; w["kernel_ms"] = out["result"].timing["kernel_b_ms"]; w["svc"] = dict(out["result"].service, launches=out["result"].timing["launches_b"])
        return w

    pool = ThreadPoolExecutor(max_workers=n_fly)

    def run(first, count):
        groups = [first + k for k in range(0, count, G)]
        chains = [[g for j, g in enumerate(groups) if j % n_fly == s] for s in range(n_fly)]
        futs = [pool.submit(lambda ch=ch, s=s: [launch(s, f0) for f0 in ch]) for s, ch in enumerate(chains)]
        return [w for f in futs for w in f.result()]

    run(0, args.warmup)
    torch.cuda.synchronize()
    t0 = time.time()
    ws = run(args.warmup, args.steps)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"{args.steps * B / dt:.1f} samples/s, {1e3 * dt / args.steps:.1f} ms per step")
    ws.sort(key=lambda w: w["t_start"])
    prev_end = None
    print("slot  asked   gate  released  (kernel)  gap_before  render_from enq_end  drained   [ms from the start of the timed region]")
    for w in ws:
        f = lambda t: 1e3 * (t - t0)
        sim_end = w["t_released"]
        r_from = w["t_start"] + w["sim_run_s"]
        gap = f(w["t_start"]) - prev_end if prev_end is not None else 0.0
        print(f"{w['slot']:4d} {f(w['t_request']):7.1f} {f(w['t_start']):7.1f} {f(sim_end):7.1f}  ({w['kernel_ms']:6.1f})  {gap:8.1f}  {f(r_from):8.1f} {f(r_from + w['render_enqueue_s']):8.1f} {f(w['t_done']):8.1f}")
        prev_end = f(sim_end)
        if w["kernel_ms"] > 500:
            print("      ^ service:", w["svc"])


if __name__ == "__main__":
    main()
