cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06
bash tools/profile_bench.sh r06 > gpurun_out/r06/profile_bench.log 2>&1
bash tools/raster_pmc.sh r06 > gpurun_out/r06/raster_pmc.log 2>&1
python tools/sim_counters.py 2>&1 | grep -v amdgpu | tail -34 > gpurun_out/r06/sim_counters.log
bash tools/sim_check.sh final2 tests 30 > gpurun_out/r06/sim_check_final2.log 2>&1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06/bench_final.json 2> gpurun_out/r06/bench_final.err
python tools/bench_line.py value value_with_csv roofline.avg_launch_ms roofline.frac roofline.traffic unet_train.value end_to_end_train.value end_to_end_gan_seg_train.value end_to_end_10k_epoch.value train_cli.value cpu_baseline.value mailbox.relaunches < gpurun_out/r06/bench_final.json
tail -3 gpurun_out/r06/sim_check_final2.log
