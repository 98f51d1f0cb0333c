#!/bin/bash
# counters of the weight-gradient kernel on one layer: L2 hit rate, fabric reads, DMA latency
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
SHAPE=${SHAPE:-"152 512 512"}
for C in TCC_HIT_sum TCC_MISS_sum FETCH_SIZE SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_EA0_RDREQ_sum SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY; do
  rm -rf /tmp/pmc_$C
  timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$C -- python tools/micro/wgrad_one.py $SHAPE > /tmp/pmc_$C.log 2>&1
  python - "$C" <<'PY'
import csv, glob, sys
c = sys.argv[1]
vals = []
for f in glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "wgrad_tr_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
            vals.append(float(r["Counter_Value"]))
print(c, "launches", len(vals), "avg", sum(vals) / max(len(vals), 1))
PY
done
