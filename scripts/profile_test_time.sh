#!/bin/bash
# Records of the test-time pipeline (run on the GPU box through gpurun from the repo root):  bash scripts/profile_test_time.sh r03
# -> gpurun_out/test_time_<tag>/: the replay / eager figure, the per-kernel averages of a rocprofv3 kernel trace of the same script,
# the RANSAC launch forms side by side, the selection modes, and the phase clocks of the hypothesis and selection kernels.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$ROOT/gpurun_out/test_time_$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 300 python3 scripts/ubench/graph_inference.py > "$OUT/graph_inference.txt" 2>/dev/null < /dev/null
timeout 300 python3 scripts/ubench/ransac_ticketed.py > "$OUT/ransac_forms.txt" 2>/dev/null < /dev/null
timeout 300 python3 scripts/ubench/select_modes.py > "$OUT/select_modes.txt" 2>/dev/null < /dev/null
timeout 300 python3 scripts/ubench/p3p_stamps.py > "$OUT/p3p_stamps.txt" 2>/dev/null < /dev/null
timeout 300 python3 scripts/ubench/sel_stamps.py > "$OUT/sel_stamps.txt" 2>/dev/null < /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tt_trace -- python3 "$ROOT/scripts/ubench/graph_inference.py" > "$OUT/trace.log" 2>&1 < /dev/null
python3 "$ROOT/scripts/ubench/kernel_avgs.py" /tmp/tt_trace 50 > "$OUT/kernel_avgs.txt" 2>&1
tail -n +1 "$OUT"/*.txt
