python -m pytest tests/test_gpu_bits.py tests/test_gpu_map_dtypes.py -x -q 2>&1 | tail -3
for v in "" binu4 binu1; do
  if [ -n "$v" ]; then export LC_AMD_LIB=$PWD/build/variants/liblc_amd_$v.so; else unset LC_AMD_LIB; fi
  echo "== variant ${v:-default}"
  bash scripts/ubench/zlmo_stream_prof.sh v2$v f16 2>&1 | grep "avg" | grep "xyz_bin_loss_bwd\|decode_gt_bwd"
done
