timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests.log 2>&1; echo "gpu tests rc=$?"; tail -5 gpurun_out/r06_gputests.log
bash scripts/ubench/zlmo_stream_prof.sh final f16 2>&1 | grep "avg"
bash scripts/ubench/zlmo_stream_prof.sh finalbf bf16 2>&1 | grep "avg"
bash scripts/ubench/zlmo_stream_prof.sh finalf32 f32 2>&1 | grep "avg"
