#!/bin/bash
# Round-1 profiling recipe (run on the GPU box through gpurun from the repo root).
# 1) kernel trace + stats of the default bench command  2) HBM traffic counters, one --pmc pass each (MI355X_MICROARCH.md
# "rocprofv3 PMC slots": FETCH_SIZE and WRITE_SIZE do not fit one pass)  3) SQ issue/wait counters for the latency-bound kernels.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${1:-r01}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 300 --warmup 30 --no-cpu-baseline --no-head"
HEAD="python3 $ROOT/bench_head.py --steps 20 --warmup 3"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o bench -- $BENCH > "$OUT/bench_trace.log" 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o bench -- $BENCH > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o bench -- $BENCH > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_sq" -o bench -- $BENCH > "$OUT/pmc_sq.log" 2>&1
if [ -f "$ROOT/bench_head.py" ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/head_trace" -o head -- $HEAD > "$OUT/head_trace.log" 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/head_pmc_fetch" -o head -- $HEAD > "$OUT/head_pmc_fetch.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/head_pmc_write" -o head -- $HEAD > "$OUT/head_pmc_write.log" 2>&1
fi
cd "$ROOT"
python3 scripts/summarize_prof.py "$OUT" > "$OUT/SUMMARY.md" 2>&1
find "$OUT" -name "*.csv" -size +2M -delete   # keep gpurun_out small: the per-dispatch traces are summarised above
ls -R "$OUT" | head -50
