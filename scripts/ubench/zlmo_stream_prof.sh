#!/bin/bash
# rocprofv3 view of scripts/ubench/zlmo_stream.py: kernel averages, then SQ counters (own pass).  bash scripts/ubench/zlmo_stream_prof.sh <tag> [dtype]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/zlmo_stream_${1:-a}
DT=${2:-f16}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/scripts/ubench/zlmo_stream.py --dtype $DT --reps 30"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o z -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d "$OUT/pmc" -o z -- $CMD > "$OUT/pmc.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)
if f:
    for r in csv.DictReader(open(f[0])):
        if "lc" in r["Name"][:70]:
            print("%-95s calls %4s avg %8.2f us" % (r["Name"][:95], r["Calls"], float(r["AverageNs"]) / 1e3))
f = glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True)
if f:
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if "lc" not in r["Kernel_Name"][:70]: continue
        acc[r["Kernel_Name"][:80]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[r["Kernel_Name"][:80]] += 1
    for k, d in acc.items():
        w = d["SQ_WAVE_CYCLES"] or 1
        print("%-80s n=%3d valu_insts/launch %.3g  VALU-active %.1f%%  wait_any %.1f%%  wait_inst %.1f%%  busy_cyc/launch %.3g" % (
            k, n[k], d["SQ_INSTS_VALU"] / n[k], 100 * d["SQ_ACTIVE_INST_VALU"] / w, 100 * d["SQ_WAIT_ANY"] / w, 100 * d["SQ_WAIT_INST_ANY"] / w, d["SQ_BUSY_CYCLES"] / n[k]))
PY
find "$OUT" -name "*.csv" -size +1500k -delete; find "$OUT" -name "*.db" -delete
