"""Record behind the mailbox protocol of the persistent simulator kernel (round 1: error bit 0x800, "a workgroup waited 30 s
for the host"). Two threads run many simulations while the main thread hammers device-wide waits.

  --park-ms 0   round 1's behaviour (wait for the answer however long it takes, here bounded by --timeout-ms): a few launches
                in several hundred fail, and the library's report shows the pattern -- the workgroup starts waiting a few
                milliseconds into the launch, the host (scanning all the time) sees the ticket only when the kernel ends,
                while the workgroup reads its own ticket back correctly;
  --park-ms 3   the shipped protocol: such a workgroup parks after 3 ms, the host serves it at the kernel boundary and
                launches again: 0 failures, `relaunches` counts the episodes.
Results: profiles/r02_mailbox_repro.log.

  python tools/repro_mailbox_deadlock.py [--park-ms 3] [--reps 300] [--iters 12,6]
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--park-ms", type=float, default=3.0)
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--timeout-ms", type=int, default=4000)
ap.add_argument("--iters", type=str, default="60,30", help="iterations of the two growth modes (short kernels = many launches)")
ap.add_argument("--pure-sync", type=int, default=1, help="main thread spins on torch.cuda.synchronize() only")
a = ap.parse_args()
os.environ["OCTA_SIM_PARK_MS"] = str(a.park_ms)
os.environ["OCTA_SIM_MAIL_TIMEOUT_MS"] = str(a.timeout_ms)

import torch
from octa_autosegmentation_amd.utils import configs
from octa_autosegmentation_amd.vessel_graph_generation import greenhouse

cfg = configs.load_generator_config()
cfg["Greenhouse"]["modes"][0]["I"], cfg["Greenhouse"]["modes"][1]["I"] = (int(v) for v in a.iters.split(","))
sims = [greenhouse.BatchSimulator(cfg, 16) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in sims]
dev = torch.cuda.current_device()
log = []


def work(slot):
    torch.cuda.set_device(dev)
    with torch.cuda.stream(streams[slot]):
        for rep in range(a.reps):
            t0 = time.time()
            try:
                res = sims[slot].run(np.arange(16) + 1000 * slot + 16 * rep)
                log.append((slot, rep, time.time() - t0, "ok", res.service))
            except Exception as e:  # noqa: BLE001
                log.append((slot, rep, time.time() - t0, "FAILED: " + str(e)[-400:], None))


ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
for t in ths:
    t.start()
n = 0
while any(t.is_alive() for t in ths):
    torch.cuda.synchronize()
    if not a.pure_sync:
        y = torch.empty(3 << 20, device="cuda"); y.fill_(1.0); del y
        torch.cuda.empty_cache()
    n += 1
for t in ths:
    t.join()
ok = [r for r in log if r[3] == "ok"]
for row in sorted(log)[:6] + [r for r in sorted(log) if r[3] != "ok"][:6]:
    print(row)
print("runs:", len(log), "failed:", len(log) - len(ok))
if ok:
    print("relaunches (episodes absorbed by parking):", sum(r[4]["relaunches"] for r in ok), " parked workgroups:", sum(r[4]["parked"] for r in ok),
          " longest pass of the service loop (ms):", max(r[4]["max_absence_ms"] for r in ok),
          " longest callback (ms):", max(r[4]["max_callback_ms"] for r in ok),
          " median run (s):", sorted(r[2] for r in ok)[len(ok) // 2])
print("device-wide waits issued by the main thread:", n)
from octa_autosegmentation_amd import _native
cnt = np.zeros(2, np.int64)
_native.lib().octa_bif_native_counts(cnt.ctypes.data)
print("bifurcation requests served natively / through the numpy callback:", cnt.tolist())
