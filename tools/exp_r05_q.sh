#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prast
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prast -- python tools/time_raster.py 128 > /dev/null 2>&1
python - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("gpurun_out/prast/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","").split("(")[0], int(r["Grid_Size_X"])*int(r["Grid_Size_Y"])))
rows.sort()
# the last label sequence: from the last raster_meta_kernel with the big grid
idx=[i for i,r in enumerate(rows) if r[2]=="raster_meta_kernel"]
a=idx[-1]
prev=None
for s,e,n,g in rows[a:a+12]:
    gap=(s-prev)/1e3 if prev else 0
    print(f"{(e-s)/1e3:9.1f} us  gap {gap:7.1f}  grid {g:10d}  {n}")
    prev=e
PY
rm -rf gpurun_out/prast
