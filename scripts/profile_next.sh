#!/bin/bash
# Kernel-trace profile of bench_next.py (the SURVEY 8f rows); run through gpurun from the repo root.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_next
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o next -- python3 $ROOT/bench_next.py > "$OUT/bench_next.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(out + "/NEXT_SUMMARY.md", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats of bench_next.py\n\n| kernel | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|\n")
    for r in rows:
        if r["Name"].startswith("lc_") or "lc::" in r["Name"]:
            o.write("| %s | %s | %s | %s | %s |\n" % (r["Name"][:80], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
print(open(out + "/NEXT_SUMMARY.md").read())
PY
find "$OUT" -name "*.csv" -size +2M -delete
