cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4/bench_final.json 2> gpurun_out/r4/bench_final.err
timeout 600 python tools/validate_many.py --digests tools/cache/oracle_digests_1000_512.npz 2>&1 | grep RESULT > gpurun_out/r4/validate.log
timeout 600 python tools/validate_many.py --digests tools/cache/oracle_digests_200000_512.npz 2>&1 | grep RESULT >> gpurun_out/r4/validate.log
cat gpurun_out/r4/validate.log
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -n 4 > gpurun_out/r4/gpu_suite.log
cat gpurun_out/r4/gpu_suite.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench_final.json').read().strip().splitlines()[-1])
r=d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'with_csv', d['value_with_csv'], 'frac', r['frac'], 'traffic', r['traffic'], 'launch', r['avg_launch_ms'])
print('serial', r['serial_depth']['per_sample_device_ms'], r['serial_depth']['per_sample_device_ms_all_slots_taken'], r['serial_depth']['frac_of_bound'])
print('raster', r['rasteriser']['label_1216']['ms_per_batch'], r['rasteriser']['image_304_x2']['ms_per_batch'], r['rasteriser']['dither_ms'])
print('voxel', r['voxeliser']['ms_per_volume'], 'G', r['gan_networks']['resnetGenerator9']['frac'], 'D', r['gan_networks']['patchGAN70x70']['frac'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['single_core_value'], 'files', d['files']['value'])
print('unet', d['unet_train']['value'], d['unet_train']['ms_per_step'], 'cli', d['train_cli']['value'], 'e2e', d['end_to_end_train']['value'], d['end_to_end_gan_seg_train']['value'])
print('cu_time', d['cu_time']['simulator_share'])
PY
