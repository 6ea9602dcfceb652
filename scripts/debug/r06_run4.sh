python tests/golden/gen_golden_ransac_f32.py > gpurun_out/gen_ransac_f32.log 2>&1; tail -3 gpurun_out/gen_ransac_f32.log
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -15 gpurun_out/r06_gputests.log
