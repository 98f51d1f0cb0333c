#!/bin/bash
# Per-launch listing of ONE steady-state U-Net training step (development aid): kernel, grid, duration, in launch order.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out
TAG=${1:-r05}
rm -rf $OUT/pstep_kt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/pstep_kt -- python tools/time_train.py ${2:-4} 1216 nchw > $OUT/${TAG}_step_kt.log 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(f"{out}/pstep_kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(int(r["Workgroup_Size_X"]), 1)))
rows.sort()
# one step = from one dice_bce_fwd to the next; take the second to last complete step
idx = [i for i, r in enumerate(rows) if "dice_bce_fwd" in r[2]]
a, b = idx[-3], idx[-2]
step = rows[a:b]
t0 = step[0][0]
with open(f"{out}/{tag}_step_launches.csv", "w") as f:
    f.write("t_start_us,dur_us,gap_before_us,workgroups,kernel\n")
    prev_end = None
    for s, e, n, wg in step:
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        f.write(f"{(s - t0) / 1e3:.1f},{(e - s) / 1e3:.1f},{gap:.1f},{wg},\"{n[:150]}\"\n")
        prev_end = e
tot = sum(e - s for s, e, _, _ in step) / 1e6
print(f"{len(step)} launches, kernel time {tot:.2f} ms, span {(step[-1][1] - step[0][0]) / 1e6:.2f} ms")
PY
rm -rf $OUT/pstep_kt
