cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 900 python tools/repro_sim_race.py 400 2>&1 | grep -v amdgpu.ids > gpurun_out/r4/repro_final2.log
tail -n 2 gpurun_out/r4/repro_final2.log
timeout 600 python tools/validate_many.py --digests tools/cache/oracle_digests_1000_512.npz 2>&1 | grep RESULT > gpurun_out/r4/validate2.log
timeout 600 python tools/validate_many.py --digests tools/cache/oracle_digests_200000_512.npz 2>&1 | grep RESULT >> gpurun_out/r4/validate2.log
cat gpurun_out/r4/validate2.log
timeout 1500 python -m pytest tests/test_sim_gpu.py tests/test_mailbox_gpu.py tests/test_fullsize_gpu.py tests/test_object_api.py -m gpu -x -q 2>&1 | tail -n 3
timeout 600 python bench.py --no-train --no-files --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'launch', d['roofline']['avg_launch_ms'], 'serial', d['roofline']['serial_depth']['per_sample_device_ms'], d['roofline']['serial_depth']['per_sample_device_ms_all_slots_taken'])
"
