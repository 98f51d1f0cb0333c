#!/bin/bash
# Steady-state kernel breakdown of the GAN-seg training step (run through gpurun from the repo root): kernel trace of
# tools/time_gan.py, aggregated over the LAST second of the run only (warm-up, MIOpen searches and allocator growth excluded).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r01}
OUT=gpurun_out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/pgan_kt -- python tools/time_gan.py 4 > $OUT/${TAG}_gan_kt.log 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(f"{out}/pgan_kt/**/*kernel_trace.csv", recursive=True):
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
end = max(r[1] for r in rows)
win = [r for r in rows if r[0] >= end - 1_000_000_000]
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in win:
    agg[n[:110]][0] += 1
    agg[n[:110]][1] += e - s
tot = sum(v[1] for v in agg.values())
with open(f"{out}/{tag}_gan_steady_kernel_stats.csv", "w") as fh:
    fh.write("Name,Calls,TotalDurationNs,Percentage\n")
    for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        fh.write(f'"{k}",{c},{d},{100 * d / tot:.2f}\n')
print(f"busy {tot / 1e6:.1f} ms of the last 1000 ms")
PY
head -45 $OUT/${TAG}_gan_steady_kernel_stats.csv
tail -2 $OUT/${TAG}_gan_kt.log
