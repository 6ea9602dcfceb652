#!/bin/bash
# Kernel-trace profile of bench_next.py (the SURVEY 8f rows); run through gpurun from the repo root.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r03}
OUT=$ROOT/gpurun_out/prof_next_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o next -- python3 $ROOT/bench_next.py > "$OUT/bench_next.log" 2>&1
# HBM traffic and issue counters, one --pmc pass each (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not share a pass)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o next -- python3 $ROOT/bench_next.py --reps 5 > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o next -- python3 $ROOT/bench_next.py --reps 5 > "$OUT/pmc_write.log" 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_sq" -o next -- python3 $ROOT/bench_next.py --reps 5 > "$OUT/pmc_sq.log" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(out + "/NEXT_SUMMARY.md", "w") as o:
    o.write("# rocprofv3 --kernel-trace --stats of bench_next.py\n\n| kernel | calls | avg ns | min ns | max ns |\n|---|---|---|---|---|\n")
    for r in rows:
        if r["Name"].startswith("lc_") or "lc::" in r["Name"] or "_ZN2lc" in r["Name"]:
            o.write("| %s | %s | %s | %s | %s |\n" % (r["Name"][:80], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"]))
    from collections import defaultdict
    for sub in ("pmc_fetch", "pmc_write", "pmc_sq"):
        for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
            acc = defaultdict(lambda: defaultdict(list))
            for r in csv.DictReader(open(f)):
                n = r["Kernel_Name"]
                if "lc_" in n:
                    n = n[n.index("lc_"):].split("(")[0].split("<")[0]
                    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
            o.write("\n## %s (mean / max per dispatch; FETCH_SIZE and WRITE_SIZE in KB, FETCH_SIZE of 16 B/lane streams reads x1/2 on gfx950)\n\n"
                    "| kernel | counter | dispatches | mean | max |\n|---|---|---|---|---|\n" % sub)
            for k, d in sorted(acc.items()):
                for c, v in sorted(d.items()):
                    o.write("| %s | %s | %d | %.4g | %.4g |\n" % (k, c, len(v), sum(v) / len(v), max(v)))
print(open(out + "/NEXT_SUMMARY.md").read())
PY
find "$OUT" -name "*.csv" -size +2M -delete
