#!/bin/bash
# on-the-fly U-Net loop: how much of the wall time has NO kernel on the GPU, and how the time splits between the simulator's launches and the rest
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/kt3
timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt3 -- python -c "
import train_synthetic
r = train_synthetic.run(steps=${STEPS:-384}, batch=4, gen_batch=512, seed0=500000, log=False, warmup=128, gan=${GAN:-False})
print(r['value'], r['ms_per_step'])
" > /tmp/kt3.log 2>&1
tail -1 /tmp/kt3.log
python - <<'PY'
import csv, glob
rows = []
for f in glob.glob("/tmp/kt3/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the timed part: from the last third of the trace on (warm-up and compilation in front)
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * 2 // 5
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
for s, e, _ in rows[1:]:
    if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
busy += cur_e - cur_s
sim = [(s, e) for s, e, n in rows if "sim_persistent" in n]
sim_t = sum(e - s for s, e in sim)
def in_sim(t): return any(s <= t < e for s, e in sim)
train = [(s, e, n) for s, e, n in rows if "sim_persistent" not in n]
k_in = sum(e - s for s, e, n in train if in_sim(s)); k_out = sum(e - s for s, e, n in train if not in_sim(s))
steps_in = sum(1 for s, e, n in train if "dice_bce_fwd" in n and in_sim(s)); steps_out = sum(1 for s, e, n in train if "dice_bce_fwd" in n and not in_sim(s))
print(f"span {span/1e6:.1f} ms, some kernel running {busy/1e6:.1f} ms ({100*busy/span:.1f} %), simulator launches {len(sim)} x {sim_t/max(len(sim),1)/1e6:.1f} ms = {100*sim_t/span:.1f} % of the span")
print(f"training steps begun inside a simulator launch {steps_in}, outside {steps_out}; ms per step inside {sim_t/1e6/max(steps_in,1):.2f}, outside {(span-sim_t)/1e6/max(steps_out,1):.2f}")
print(f"other kernels' time inside launches {k_in/1e6:.1f} ms, outside {k_out/1e6:.1f} ms")
PY
