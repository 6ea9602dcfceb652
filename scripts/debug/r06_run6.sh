python -m pytest tests/test_gpu_bits.py tests/test_gpu_dense.py tests/test_gpu_map_dtypes.py tests/test_gpu_xyz_bin_sharded.py tests/test_gpu_train_step.py -x -q 2>&1 | tail -8
bash scripts/ubench/zlmo_stream_prof.sh v1 f16 2>&1 | tail -20
