cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
for n in "$@"; do
  echo "=== $n"
  OCTA_HIP_LIB=$GRAFT_REPO_ROOT/gpurun_variants/liboctahip_$n.so timeout 300 python tools/sim_phases.py 512 2 2>&1 | grep -v amdgpu.ids | tail -n 3 | tee gpurun_out/r4/phases512_$n.log
done
