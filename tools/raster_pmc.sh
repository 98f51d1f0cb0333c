#!/bin/bash
# HBM-side traffic of the rasteriser's kernels (run through gpurun from the repo root): FETCH_SIZE and WRITE_SIZE, one counter per pass,
# over `bench.py --pmc-child --batch 128` (two generate() calls of 128 triples: 2 x 128 images at 304^2 + 128 labels at 1216^2 + dither).
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r05}
OUT=gpurun_out
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/praster_$C
  timeout 600 rocprofv3 --pmc $C --output-format csv -d $OUT/praster_$C -- python bench.py --pmc-child --batch 128 > $OUT/${TAG}_raster_pmc_$C.log 2>&1
done
python - "$OUT" "$TAG" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"{out}/praster_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != c:
                continue
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
            if "raster" in k or "dither" in k or "read_back" in k or "max_u8" in k:
                if "render" in k:
                    k += " [label 1216^2]" if int(r["Grid_Size"]) > 100000000 // 4 else " [images 304^2]"
                agg[k][c] += float(r["Counter_Value"]) / 2.0          # two generate() calls
                if c == "FETCH_SIZE":
                    agg[k]["n"] += 0.5
lines = ["kernel,launches_per_128,fetch_MB(2x FETCH_SIZE),write_MB,total_MB"]
tot_w = tot = 0.0
for k, v in sorted(agg.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])):
    f, w = 2 * v["FETCH_SIZE"] / 1024, v["WRITE_SIZE"] / 1024
    tot_w += w; tot += f + w
    lines.append(f'"{k}",{v["n"]:.1f},{f:.1f},{w:.1f},{f + w:.1f}')
outb = 128 * (1216 * 1216 * 2 + 3 * 304 * 304) / 2**20
lines.append(f'TOTAL,,,{tot_w:.1f},{tot:.1f}')
lines.append(f'output bytes (grey label + binarised label + two 304^2 rasters + their maximum) MB,,,{outb:.1f},')
lines.append(f'write amplification (WRITE / output),,,{tot_w / outb:.2f},')
open(f"{out}/{tag}_raster_pmc.log", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/praster_FETCH_SIZE $OUT/praster_WRITE_SIZE
