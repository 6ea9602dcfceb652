#!/bin/bash
# Copy the judged summaries of the last scripts/profile_round.sh run from gpurun_out/ (scratch) into profiles/<round>/ (tracked).
#   bash scripts/collect_profiles.sh r02 <git sha the run was taken at>
set -eu
TAG=${1:-r03}
SHA=${2:-unknown}
SRC=gpurun_out/prof_$TAG
DST=profiles/$TAG
mkdir -p "$DST/trace" "$DST/head_trace"
cp "$SRC/SUMMARY.md" "$SRC/pmc_traffic.json" "$SRC/sweep_pose_unit.json" "$DST/"
cp "$SRC"/trace/*kernel_stats.csv "$SRC"/trace/*domain_stats.csv "$DST/trace/" 2>/dev/null || true
cp "$SRC"/head_trace/*kernel_stats.csv "$DST/head_trace/" 2>/dev/null || true
grep -v amdgpu "$SRC/sweep_events.jsonl" > "$DST/sweep_events.jsonl" || true
for f in bench_default bench_k20 bench_head_f32 bench_head_bf16 bench_head_f16; do
  [ -s "$SRC/$f.json" ] && python3 -c "import json; json.dump(json.load(open('$SRC/$f.json')), open('$DST/$f.json', 'w'), indent=1)"
done
[ -s "$SRC/fuzz_parity_400.txt" ] && { echo "# tests/fuzz_parity.py --cases 400 --seed 7 on one MI355X, commit $SHA (the round-1 record of the same sweep: profiles/r01/fuzz_parity.txt)"; cat "$SRC/fuzz_parity_400.txt"; } > "$DST/fuzz_parity_400.txt"
[ -s "$SRC/pnp_iter_clock.txt" ] && { echo "# scripts/ubench/pnp_iter_clock.py (diagnostic build -DLC_TRACE_CLOCK of commit $SHA; shader cycles, B=256 N=64)"; cat "$SRC/pnp_iter_clock.txt"; } > "$DST/pnp_iter_clock.txt"
[ -s "$SRC/diag_stamps.txt" ] && { echo "# scripts/diag_stamps.py (diagnostic build -DLC_STAMPS of commit $SHA: s_memtime stamps per phase, shader cycles; the stamps themselves cost ~11 %)"; cat "$SRC/diag_stamps.txt"; } > "$DST/diag_stamps.txt"
[ -s "$SRC/lockstep.txt" ] && { echo "# scripts/pnp_numerics/lockstep_report.py on one MI355X (commit $SHA): the kernel's trust-region schedule vs the oracle's, per problem set of tests/pnp_cases.py"; echo "# 'lock-step' = identical accept/reject/invalid/tolerance sequence over the whole solve; then the max relative difference of every traced column over those jobs"; cat "$SRC/lockstep.txt"; } > "$DST/pnp_lockstep.txt"
TT=gpurun_out/test_time_$TAG
if [ -d "$TT" ]; then  # scripts/profile_test_time.sh
  mkdir -p "$DST/test_time"
  for f in zlmo glmo kernel_avgs_zlmo kernel_avgs_glmo graph_inference_hybrid select_stamps graph_inference kernel_avgs ransac_forms select_modes p3p_stamps sel_stamps; do
    [ -s "$TT/$f.txt" ] && { echo "# scripts/profile_test_time.sh on one MI355X, commit $SHA"; cat "$TT/$f.txt"; } > "$DST/test_time/$f.txt"
  done
fi
NX=gpurun_out/prof_next_$TAG
if [ -d "$NX" ]; then  # scripts/profile_next.sh
  mkdir -p "$DST/next"
  cp "$NX/NEXT_SUMMARY.md" "$DST/next/" 2>/dev/null || true
  cp "$NX"/trace/*kernel_stats.csv "$DST/next/next_kernel_stats.csv" 2>/dev/null || true
  [ -s "$NX/bench_next.jsonl" ] && grep '^{' "$NX/bench_next.jsonl" > "$DST/next/bench_next.jsonl"
fi
ls -la "$DST"
