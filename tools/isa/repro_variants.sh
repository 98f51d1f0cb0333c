cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
REPS=$1; shift
for n in "$@"; do
  echo "=== $n"
  OCTA_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_variants/liboctahip_$n.so timeout 600 python tools/repro_sim_race.py $REPS 2>&1 | grep -v amdgpu.ids | cut -c1-250 > gpurun_out/r4/repro_$n.log
  tail -n 1 gpurun_out/r4/repro_$n.log
done
