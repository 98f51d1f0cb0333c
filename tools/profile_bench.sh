#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the default bench command        -> gpurun_out/r03_bench_kernel_stats.csv
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE, one counter per pass, launches of the default shape (4 x 128 samples; two in flight, one persistent kernel at a time)
#      -> gpurun_out/r03_bench_pmc_summary.csv
# (counter passes never combined with tracing: see the task's profiling rules)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r06}
OUT=gpurun_out
mkdir -p $OUT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_kt -- python bench.py --no-cpu-baseline --no-pmc --no-long > $OUT/${TAG}_bench_stdout.log 2>&1
find $OUT/prof_kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_bench_kernel_stats.csv
# the stats file averages launches of every size (512-sample launches of the headline leg, 128 / 256-sample launches of the solo, files and
# on-the-fly legs): the per-size averages of the persistent kernel, from the raw trace, are what bench.py's roofline.avg_launch_ms (HIP
# events around the 512-sample launches) has to agree with
python - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
rows = []
for f in glob.glob(f"{out}/prof_kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sim_persistent_kernel" in r["Kernel_Name"]:
            wgs = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
            rows.append((int(r["Start_Timestamp"]), wgs, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
rows.sort()
# bench.py's headline leg runs first: with the default --warmup 4 --steps 8 --group 4 its launches are the first three of 512 workgroups
# (one warm-up launch, two timed ones); later 512-workgroup launches belong to the generator CLI of the files leg
head = [d for _, w, d in rows if w == 512][:3]
agg = collections.defaultdict(list)
for _, w, d in rows:
    agg[w].append(d)
with open(f"{out}/{tag}_bench_sim_launches.csv", "w") as f:
    f.write("kernel,launches_of,launches,avg_ms,min_ms,max_ms\n")
    if head:
        f.write("sim_persistent_kernel,512 workgroups: headline leg (first 3),%d,%.3f,%.3f,%.3f\n" % (len(head), sum(head) / len(head), min(head), max(head)))
    for wgs, v in sorted(agg.items(), reverse=True):
        f.write("sim_persistent_kernel,%d workgroups: all legs,%d,%.3f,%.3f,%.3f\n" % (wgs, len(v), sum(v) / len(v), min(v), max(v)))
PY
rm -rf $OUT/prof_kt   # the raw trace is large; only the summaries are kept
grep '"metric"' $OUT/${TAG}_bench_stdout.log | tail -1 > $OUT/${TAG}_bench_line_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $C --output-format csv -d $OUT/prof_$C -- python bench.py --steps 8 --warmup 4 --no-cpu-baseline --no-train --no-files --no-pmc --no-long > $OUT/${TAG}_pmc_$C.log 2>&1
done
python - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
rows = []
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{out}/prof_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != c:
                continue
            k = r["Kernel_Name"].replace(",", ";")[:90] + f" [grid {r['Grid_Size']}]"     # launches of different sizes are different rows
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        rows.append((c, k, n, round(v, 1), round(v / max(n, 1), 1)))
with open(f"{out}/{tag}_bench_pmc_summary.csv", "w") as f:
    f.write("counter,kernel,launches,sum_KB,avg_KB_per_launch\n")
    for r in rows:
        f.write('%s,"%s",%d,%s,%s\n' % r)
print("pmc rows:", len(rows))
PY
head -12 $OUT/${TAG}_bench_kernel_stats.csv | cut -c1-200
head -8 $OUT/${TAG}_bench_pmc_summary.csv
cut -c1-300 $OUT/${TAG}_bench_line_under_rocprof.json
