#!/bin/bash
# SQ counters of the rasteriser's render kernel (labels): issue mix, lane use, waiting
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for C in SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAVES; do
  rm -rf /tmp/pmc_$C
  timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -- python tools/time_raster.py > /tmp/pmc_$C.log 2>&1
  python - "$C" <<'PY'
import csv, glob, sys
c = sys.argv[1]
vals = []
for f in glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "raster_render_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c and int(r["Grid_Size"]) > 40000000:
            vals.append(float(r["Counter_Value"]))
print(c, len(vals), sum(vals) / max(len(vals), 1))
PY
done
