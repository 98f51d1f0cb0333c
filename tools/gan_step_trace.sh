#!/bin/bash
# Per-launch listing of ONE steady-state GAN-seg training step (development aid): kernel, workgroups, duration, in launch order
# -> gpurun_out/<tag>_gan_step_launches.csv, plus the step's kernel time by kernel name
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
TAG=${1:-r05}
rm -rf $OUT/gstep_kt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/gstep_kt -- python tools/time_gan.py ${2:-4} > $OUT/${TAG}_gan_step_kt.log 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(f"{out}/gstep_kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]), 1)))
rows.sort()
# one step = from one weight pack of the generator's optimiser step to the next: the pack kernel runs a fixed number of times per step;
# take the launches between the (k)th and (k + per_step)th pack kernels near the end
idx = [i for i, r in enumerate(rows) if "pack_weights_kernel" in r[2]]
per_step = 3 if len(idx) % 3 == 0 else (2 if len(idx) % 2 == 0 else 1)
for cand in (3, 2, 4, 1):
    if len(idx) % cand == 0 and len(idx) // cand >= 13:
        per_step = cand
        break
a, b = idx[-2 * per_step], idx[-per_step]
step = rows[a:b]
t0 = step[0][0]
agg = {}
with open(f"{out}/{tag}_gan_step_launches.csv", "w") as f:
    f.write("t_start_us,dur_us,gap_before_us,workgroups,kernel\n")
    prev_end = None
    for s, e, n, wg in step:
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        f.write(f"{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{gap:.1f},{wg},\"{n[:150]}\"\n")
        prev_end = max(prev_end or 0, e)
        k = n.split("(")[0][-60:]
        agg[k] = agg.get(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += (e - s) / 1e3
tot = sum(e - s for s, e, _, _ in step) / 1e6
print(f"{len(step)} launches ({per_step} weight packs per step), kernel time {tot:.2f} ms, span {(step[-1][1] - step[0][0]) / 1e6:.2f} ms")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t:9.1f} us {c:4d} x  {k}")
PY
rm -rf $OUT/gstep_kt
