#!/bin/bash
# the headline leg's slow mode (kernels of ~520 ms): what is on the GPU around the START of each persistent kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
rm -rf /tmp/ktc; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktc -- python bench.py --steps 20 --warmup 5 --no-train --no-files --no-long --no-cpu-baseline --no-pmc > /tmp/ktc.log 2>/dev/null
python - <<'PY'
import csv, glob, re, json
line = [l for l in open("/tmp/ktc.log") if l.startswith("{")]
if line:
    d = json.loads(line[-1]); print("RUN", round(d["value"], 1), "samples/s, kernel", round(d["slot_cycle"]["kernel_ms"], 1))
rows = []
for f in glob.glob("/tmp/ktc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][-34:], r.get("Stream_Id", r.get("Queue_Id", "?"))))
rows.sort()
sims = [r for r in rows if "sim_persistent_kernel" in r[2]]
prev_end = None
for s in sims:
    near = [r for r in rows if r is not s and r[1] > s[0] - 2e6 and r[0] < s[0] + 80e6 and "sim_persistent" not in r[2]]
    agg = {}
    for a, b, n, q in near:
        e = agg.setdefault(n, [0, 0.0, 1e18, -1e18]); e[0] += 1; e[1] += (b - a) / 1e6; e[2] = min(e[2], (a - s[0]) / 1e6); e[3] = max(e[3], (b - s[0]) / 1e6)
    gap = (s[0] - prev_end) / 1e6 if prev_end else 0
    print(f"kernel {(s[1]-s[0])/1e6:6.1f} ms  gap before {gap:6.1f}  | " + "; ".join(f"{n} x{c} {t:.1f}ms [{lo:+.1f},{hi:+.1f}]" for n, (c, t, lo, hi) in sorted(agg.items(), key=lambda kv: kv[1][2])[:7]))
    prev_end = s[1]
PY
done
