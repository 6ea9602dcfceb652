#!/bin/bash
# Per-round profiling recipe (run on the GPU box through gpurun from the repo root):
#   bash scripts/profile_round.sh r02 <git sha of the build>
# 1) kernel trace + stats of the default bench command  2) HBM traffic counters, one --pmc pass each (MI355X_MICROARCH.md
# "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one pass)  3) SQ issue/wait counters of the pose-unit kernels
# 4) the same for the keypoint head  5) the batch-size sweep of the pose unit (kernel trace + SQ counters, grouped by grid size).
# --pmc passes carry --kernel-trace only (gpurun refuses --pmc with other trace domains); the program follows `--` directly.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
SHA=${2:-unknown}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 300 --warmup 30 --regions 3 --steady-batch 0 --no-cpu-baseline --no-head"
HEAD="python3 $ROOT/bench_head.py --steps 20 --warmup 3"
SWEEP="python3 $ROOT/scripts/ubench/sweep_pose_unit.py"
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $BENCH > "$OUT/bench_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- $BENCH > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$OUT/pmc_sq" -o bench -- $BENCH > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/head_trace" -o head -- $HEAD > "$OUT/head_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/head_pmc_fetch" -o head -- $HEAD > "$OUT/head_pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/head_pmc_write" -o head -- $HEAD > "$OUT/head_pmc_write.log" 2>&1
$SWEEP > "$OUT/sweep_events.jsonl" 2> "$OUT/sweep_events.err"
LC_SWEEP_REPS=10 rocprofv3 --kernel-trace --output-format csv -d "$OUT/sweep_trace" -o sweep -- $SWEEP > "$OUT/sweep_trace.log" 2>&1
LC_SWEEP_REPS=6 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d "$OUT/sweep_pmc_sq" -o sweep -- $SWEEP > "$OUT/sweep_pmc_sq.log" 2>&1
cd "$ROOT"
python3 scripts/summarize_prof.py "$OUT" "$SHA" "$BENCH" > "$OUT/SUMMARY.md" 2>&1
find "$OUT" -name "*.csv" -size +2M -delete   # keep gpurun_out small: the per-dispatch traces are summarised above
find "$OUT" -name "*.db" -delete
ls -R "$OUT" | head -60
