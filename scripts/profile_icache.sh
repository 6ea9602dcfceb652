#!/bin/bash
# Instruction-fetch counters of the three large pose kernels (run on the GPU box through gpurun from the repo root):
#   bash scripts/profile_icache.sh r04
# lc_pnp_lm_chain_kernel (71 KB of code), lc_pose_unit_dense_kernel<8> (84 KB) against lc_pose_unit_kernel<1> (38 KB): does the kernel
# outgrow the 64 KB instruction cache a pair of compute units shares?  One --pmc pass per counter group (--kernel-trace only).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
OUT=$ROOT/gpurun_out/icache_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -E "SQC_ICACHE|SQ_IFETCH|SQ_INST_CYCLES_VMEM|SQC_INST|ICACHE|IFETCH|SQ_WAIT_INST|SQ_INST_LEVEL" | sort -u | head -60 > "$OUT/counters_listed.txt" 2>&1
BENCH="python3 $ROOT/bench.py --steps 100 --warmup 10 --regions 3 --steady-batch 0 --no-cpu-baseline --no-head"
export CFG=glmo
TT="python3 $ROOT/scripts/ubench/config_test_time.py"
i=0
for GROUP in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVES"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d "$OUT/bench_g$i" -o bench -- $BENCH > "$OUT/bench_g$i.log" 2>&1
    rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d "$OUT/tt_g$i" -o tt -- $TT > "$OUT/tt_g$i.log" 2>&1
done
cd "$ROOT"
python3 scripts/summarize_icache.py "$OUT" > "$OUT/ICACHE.md" 2>&1
find "$OUT" -name "*.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete
cat "$OUT/counters_listed.txt" "$OUT/ICACHE.md"
