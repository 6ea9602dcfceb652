#!/bin/bash
# Records of the test-time pipeline at the reference's own configs (run on the GPU box through gpurun from the repo root):
#   bash scripts/profile_test_time.sh r04
# -> gpurun_out/test_time_<tag>/: per config (zlmo, glmo) the replay / eager figure and the per-kernel averages of a rocprofv3 kernel trace of
# the same script; plus the RANSAC launch forms side by side, the selection modes and the phase clocks of the hypothesis / selection kernels.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
OUT=$ROOT/gpurun_out/test_time_$TAG
mkdir -p "$OUT"
cd "$ROOT"
for CFG in zlmo glmo; do
    export CFG
    timeout 300 python3 scripts/ubench/config_test_time.py > "$OUT/$CFG.txt" 2>/dev/null < /dev/null
done
if [ "${2:-}" = "all" ]; then
    timeout 300 python3 scripts/ubench/graph_inference.py > "$OUT/graph_inference_hybrid.txt" 2>/dev/null < /dev/null
    timeout 300 python3 scripts/ubench/ransac_ticketed.py > "$OUT/ransac_forms.txt" 2>/dev/null < /dev/null
    timeout 300 python3 scripts/ubench/select_modes.py > "$OUT/select_modes.txt" 2>/dev/null < /dev/null
    timeout 300 python3 scripts/ubench/p3p_stamps.py > "$OUT/p3p_stamps.txt" 2>/dev/null < /dev/null
    timeout 300 python3 scripts/ubench/sel_stamps.py > "$OUT/sel_stamps.txt" 2>/dev/null < /dev/null
fi
cd /tmp && export TMPDIR=/tmp
for CFG in zlmo glmo; do
    export CFG
    rm -rf /tmp/tt_trace_$CFG
    timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tt_trace_$CFG -- python3 "$ROOT/scripts/ubench/config_test_time.py" > "$OUT/trace_$CFG.log" 2>&1 < /dev/null
    python3 "$ROOT/scripts/ubench/kernel_avgs.py" /tmp/tt_trace_$CFG 50 > "$OUT/kernel_avgs_$CFG.txt" 2>&1
done
tail -n +1 "$OUT"/*.txt
