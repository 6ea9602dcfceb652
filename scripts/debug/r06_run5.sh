python -m pytest tests/test_gpu_ddp_ranks.py -q -k "comm_report" 2>&1 | tail -5
bash scripts/profile_g1.sh r06 > gpurun_out/g1_r06.log 2>&1; sed -n 1,12p gpurun_out/g1_r06/G1_SUMMARY.md
