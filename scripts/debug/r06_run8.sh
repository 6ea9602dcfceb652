python -m pytest tests/test_gpu_bits.py tests/test_gpu_map_dtypes.py tests/test_gpu_dense.py tests/test_gpu_xyz_bin_sharded.py -x -q 2>&1 | tail -3
bash scripts/ubench/zlmo_stream_prof.sh v3 f16 2>&1 | grep "avg\|xyz_bin_loss_fwd\|bwd_tile"
