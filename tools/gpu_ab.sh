#!/bin/bash
# ONE parameterised A/B runner for `gpurun` (round 6: replaces the 46 one-letter exp_r05_*.sh scripts and this round's exp_r06_*.sh).
#
#   tools/gpu_ab.sh TAG REPS 'label::ENV=.. ENV2=..::command' ['label2::::command2' ...]
#
# Runs every variant REPS times, ALTERNATING the variants inside a repetition (box-to-box and minute-to-minute drift hits both sides), and
# appends "label rep k: <last line of the command's stdout>" to gpurun_out/ab/TAG.log. The command's stderr goes to gpurun_out/ab/TAG.err.
# Examples (the round-6 measurements of DESIGN.md, each one `gpurun -- 'bash tools/gpu_ab.sh ...'`):
#   headline leg five times:      tools/gpu_ab.sh order 5 'gated::::python bench.py --no-train --no-files --no-pmc --no-cpu-baseline --steps 20 --warmup 5 | python tools/bench_line.py value slot_cycle.render_enqueue_ms cu_time.simulator_share'
#   a library variant against the shipped one (tools/build_sim_variant.py NAME -DFLAG):
#                                 tools/gpu_ab.sh seqprof 1 'shipped::::python tools/sim_phases.py 512 1' 'variant::OCTA_HIP_LIB=$PWD/gpurun_variants/liboctahip_NAME.so::python tools/sim_phases.py 512 1'
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}" || exit 1
TAG=$1; REPS=$2; shift 2
mkdir -p gpurun_out/ab
export PYTHONUNBUFFERED=1
for rep in $(seq 1 "$REPS"); do
  for spec in "$@"; do
    label=${spec%%::*}; rest=${spec#*::}; envs=${rest%%::*}; cmd=${rest#*::}
    envs=$(eval echo "$envs")
    out=$(env $envs bash -c "$cmd" 2>>gpurun_out/ab/"$TAG".err | grep -v "amdgpu.ids" | tail -n 1)
    echo "$label rep $rep: $out" | tee -a gpurun_out/ab/"$TAG".log
  done
done
