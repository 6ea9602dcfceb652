for v in "" fskip; do
  if [ -n "$v" ]; then export LC_AMD_LIB=$PWD/build/variants/liblc_amd_$v.so; else unset LC_AMD_LIB; fi
  echo "== variant ${v:-default}"
  bash scripts/ubench/zlmo_stream_prof.sh v13$v f16 2>&1 | grep "xyz_bin_loss_fwd"
done
