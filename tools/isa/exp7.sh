cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
bash tools/isa/repro_variants.sh 30 vT3
echo "=== vI one workgroup per CU"
OCTA_SIM_GRID=256 OCTA_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_variants/liboctahip_vI.so timeout 600 python tools/repro_sim_race.py 20 2>&1 | tail -n 1
