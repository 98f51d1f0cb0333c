cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
L=$GRAFT_REPO_ROOT/gpurun_variants/liboctahip_vDS.so
for mode in default finegrained uncached; do
  echo "=== ds build, OCTA_SIM_ALLOC=$mode"
  OCTA_SIM_ALLOC=$mode OCTA_HIP_LIB=$L timeout 600 python tools/repro_sim_race.py 30 2>&1 | tail -n 1
done
