cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
for rep in 1 2; do
for v in "" "OCTA_EPI_STATS=1" "OCTA_LAZY_NORM=1" "OCTA_EPI_STATS=1 OCTA_LAZY_NORM=1"; do
  echo "== rep $rep [$v]"
  env $v timeout 300 python tools/time_train.py 4 1216 2>&1 | grep "DynUNet train"
done
done | tee gpurun_out/r4/norm_ab.log
