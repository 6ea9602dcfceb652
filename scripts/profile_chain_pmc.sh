#!/bin/bash
# SQ counters of the test-time chain's kernels at a reference config (run on the GPU box through gpurun from the repo root):
#   bash scripts/profile_chain_pmc.sh r04 zlmo
# One --pmc pass per counter group (--kernel-trace only), per-kernel means by scripts/summarize_chain_pmc.py.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
export CFG=${2:-zlmo}
OUT=$ROOT/gpurun_out/chain_pmc_${TAG}_$CFG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
TT="python3 $ROOT/${PMC_SCRIPT:-scripts/ubench/config_test_time.py}"  # PMC_SCRIPT=bench_next.py: the f-row kernels instead of the chain
i=0
for GROUP in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT"; do
    i=$((i + 1))
    rocprofv3 --kernel-trace --pmc $GROUP --output-format csv -d "$OUT/g$i" -o tt -- $TT > "$OUT/g$i.log" 2>&1
done
cd "$ROOT"
python3 scripts/summarize_chain_pmc.py "$OUT" > "$OUT/CHAIN_PMC.md" 2>&1
find "$OUT" -name "*.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete
cat "$OUT/CHAIN_PMC.md"
