cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
V=$GRAFT_REPO_ROOT/gpurun_variants
for n in unii_d1 unii_d2 unii_d3; do
  OCTA_HIP_LIB=$V/liboctahip_$n.so timeout 500 python tools/repro_sim_race.py 150 > gpurun_out/r4/exp2_$n.log 2>&1
  tail -n 3 gpurun_out/r4/exp2_$n.log
done
