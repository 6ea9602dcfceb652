python -m pytest tests/test_gpu_bits.py tests/test_gpu_dense.py tests/test_gpu_xyz_bin_sharded.py -x -q 2>&1 | tail -4
