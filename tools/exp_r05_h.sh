#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r05_exp_h.log
: > $L
run() { echo "### $*" >> $L; ( env "$@" ) 2>&1 | grep -E "DynUNet train|GAN-seg step|Error|error" >> $L; }
run OCTA_S2T=1 python tools/time_train.py 4
run OCTA_S2T=1 OCTA_S2T_BN=32 python tools/time_train.py 4
run OCTA_S2T=1 python tools/time_train.py 4
run OCTA_S2T=1 OCTA_S2T_BN=32 python tools/time_train.py 4
cat $L
OCTA_S2T_BN=32 bash tools/step_trace.sh r05_bn32 | tail -1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r05_bn32_step_launches.csv')))
for r in rows:
    if 's2t' in r['kernel']: print(r['dur_us'], r['workgroups'], r['kernel'][30:60])
PY
