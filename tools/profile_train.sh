#!/bin/bash
# rocprofv3 evidence for the U-Net training step on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats, then ONE counter per pass (never combined with tracing): MFMA busy cycles, LDS instructions / active cycles /
#   bank conflicts, VALU / SALU instruction counts, wait cycles, FETCH_SIZE and WRITE_SIZE.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r01}
OUT=gpurun_out
CMD="python tools/time_train.py 4 1216"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ptrain_kt -- $CMD > $OUT/${TAG}_train_kt.log 2>&1
find $OUT/ptrain_kt -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/${TAG}_train_mfma_kernel_stats.csv
COUNTERS=${COUNTERS:-"SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU FETCH_SIZE WRITE_SIZE"}
for C in $COUNTERS; do
  timeout 900 rocprofv3 --pmc $C --output-format csv -d $OUT/ptrain_$C -- $CMD > $OUT/${TAG}_train_pmc_$C.log 2>&1
done
export COUNTERS
python - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
dur = {}
for r in csv.DictReader(open(f"{out}/{tag}_train_mfma_kernel_stats.csv")):
    dur[r["Name"][:70]] = (int(r["Calls"]), float(r["TotalDurationNs"]))
rows = []
import os
for c in os.environ.get("COUNTERS", "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU FETCH_SIZE WRITE_SIZE").split():
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{out}/ptrain_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != c:
                continue
            k = r["Kernel_Name"][:70]
            agg[k][0] += 1
            agg[k][1] += float(r["Counter_Value"])
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:12]:
        calls, tot_ns = dur.get(k, (0, 0.0))
        rows.append((c, k.replace(",", ";"), n, v, v / max(n, 1), tot_ns / max(calls, 1)))
with open(f"{out}/{tag}_train_pmc_summary.csv", "w") as f:
    f.write("counter,kernel,launches,sum,avg_per_launch,avg_kernel_ns_from_trace\n")
    for r in rows:
        f.write('%s,"%s",%d,%.1f,%.1f,%.1f\n' % r)
print(open(f"{out}/{tag}_train_pmc_summary.csv").read()[:3000])
PY
