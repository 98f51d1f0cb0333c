#!/bin/bash
# GAN-seg step: per-launch durations of the norm kernels by grid size (are the small planes launch-bound?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/ktg; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktg -- python tools/time_gan.py 4 > /tmp/ktg.log 2>&1
python - <<'PY'
import csv, glob, collections, re
rows = []
for f in glob.glob("/tmp/ktg/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"])))
rows.sort()
end = rows[-1][1]
win = [r for r in rows if r[0] >= end - 480_000_000]           # ~10 steps
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n, gx, gy in win:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n).split("(")[0]
    if n.startswith("in_nhwc") or "thin" in n or "blur" in n or "reflect" in n:
        agg[(n[:40], gx, gy)][0] += 1; agg[(n[:40], gx, gy)][1] += e - s
tot = sum(e - s for s, e, *_ in win)
print("window kernel time ms", tot / 1e6)
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{t/1e6:8.2f} ms {c:5d} calls avg {t/c/1e3:7.1f} us  grid {k[1]}x{k[2]}  {k[0]}")
PY
