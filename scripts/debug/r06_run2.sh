python -m pytest tests/test_gpu_xyz_bin_sharded.py -q 2>&1 | tail -3
bash scripts/profile_g1.sh r06 > gpurun_out/g1_r06.log 2>&1; tail -30 gpurun_out/g1_r06.log
python -m pytest tests/test_gpu_train_step.py tests/test_gpu_ddp_ranks.py -q -k "zlmo" 2>&1 | tail -5
