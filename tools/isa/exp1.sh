set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
V=$GRAFT_REPO_ROOT/gpurun_variants
OCTA_HIP_LIB=$V/liboctahip_unii_dbg.so timeout 600 python tools/repro_sim_race.py 100 > gpurun_out/r4/exp1_unii_dbg.log 2>&1
OCTA_HIP_LIB=$V/liboctahip_unii.so timeout 400 python tools/repro_sim_race.py 100 > gpurun_out/r4/exp1_unii.log 2>&1
OCTA_SIM_GRID=256 OCTA_HIP_LIB=$V/liboctahip_unii.so timeout 400 python tools/repro_sim_race.py 60 > gpurun_out/r4/exp1_unii_grid256.log 2>&1
tail -5 gpurun_out/r4/exp1_*.log
