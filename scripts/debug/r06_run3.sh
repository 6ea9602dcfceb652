python -m pytest tests/test_gpu_xyz_bin_sharded.py -q -x -k "sharing" 2>&1 | grep -v Gloo | grep -B2 -A25 "Error" | head -60
