cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONUNBUFFERED=1
timeout 1200 python -m pytest tests/test_cli_gpu.py tests/test_seams.py tests/test_training_cli_gpu.py tests/test_postproc_gpu.py -m gpu -x -q 2>&1 | tail -n 6
timeout 1500 python bench.py > gpurun_out/r4/bench2.json 2> gpurun_out/r4/bench2.err
tail -c 600 gpurun_out/r4/bench2.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4/bench2.json').read().strip().splitlines()[-1])
print('value', d['value'], 'value_with_csv', d.get('value_with_csv'), 'rccl_ranks', d.get('rccl_ranks'))
print('voxeliser', d['roofline'].get('voxeliser'))
print('gan', d['roofline'].get('gan_networks'))
print('unet', d['unet_train']['value'], 'cli', d['train_cli']['value'], 'e2e', d['end_to_end_train']['value'], d['end_to_end_gan_seg_train']['value'])
PY
