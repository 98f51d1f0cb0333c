#!/bin/bash
# usage: tools/sim_variants_ab.sh TAG REPS NAME [NAME ...]   -- tools/sim_phases.py 512 1 of the shipped library and of gpurun_variants/liboctahip_NAME.so, alternating
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
TAG=$1; REPS=$2; shift 2
specs=('shipped::::python tools/sim_phases.py 512 1 2>&1 | grep -A2 "^rep 0" | tr "\n" " " | cut -c1-460')
for n in "$@"; do specs+=("$n::OCTA_HIP_LIB=\$PWD/gpurun_variants/liboctahip_$n.so::python tools/sim_phases.py 512 1 2>&1 | grep -A2 \"^rep 0\" | tr \"\\n\" \" \" | cut -c1-460"); done
bash tools/gpu_ab.sh "$TAG" "$REPS" "${specs[@]}"
