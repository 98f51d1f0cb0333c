#!/bin/bash
# code size of the simulator's phases (bytes of gfx950 code), compiled one kernel per phase: CPU only, ~2 min
# usage: tools/sim_code_size.sh            (needs hipcc; writes nothing into the tree)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/octa_codesize.XXXX)
cd "$T"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DOCTA_SIM_CODE_SIZE_PROBE -c "$ROOT/octa_autosegmentation_amd/csrc/sim.hip" -o probe.o --save-temps -I"$ROOT/octa_autosegmentation_amd/csrc" 2>/dev/null
python3 - <<'PY'
import re
name = None
for line in open("sim-hip-amdgcn-amd-amdhsa-gfx950.s"):
    m = re.match(r"^(_ZN\S+|\w+):\s+; @", line)
    if m: name = m.group(1)
    m = re.match(r"; codeLenInByte = (\d+)", line)
    if m and name:
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
        short = re.sub(r"E(NS_|PK|Pd|ii|i).*$", "", short)
        print(f"{int(m.group(1)):9d}  {short}")
        name = None
PY
rm -rf "$T"
