#!/bin/bash
# what happens between two persistent-kernel launches of the headline leg
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/ktb; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktb -- python bench.py ${BENCH_ARGS:---steps 20 --warmup 5} --no-train --no-files --no-long --no-cpu-baseline --no-pmc > /tmp/ktb.log 2>&1
python - <<'PY'
import csv, glob, re
rows = []
for f in glob.glob("/tmp/ktb/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:40]))
rows.sort()
sims = [r for r in rows if "sim_persistent_kernel" in r[2]]
for a, b in zip(sims[:-1], sims[1:]):
    gap = (b[0] - a[1]) / 1e6
    inside = [r for r in rows if r[0] >= a[1] and r[1] <= b[0]]
    agg = {}
    for s, e, n in inside: agg[n] = agg.get(n, 0) + (e - s) / 1e6
    top = sorted(agg.items(), key=lambda kv: -kv[1])[:4]
    print(f"kernel {(a[1]-a[0])/1e6:7.1f} ms, gap to next {gap:7.2f} ms; in the gap: " + ", ".join(f"{n} {t:.1f}" for n, t in top))
PY
