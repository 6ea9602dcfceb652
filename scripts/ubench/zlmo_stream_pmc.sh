#!/bin/bash
# HBM-side counters of scripts/ubench/zlmo_stream.py (the streaming lc_* kernels at zlmo's training shape, cold inputs): FETCH_SIZE and WRITE_SIZE, one --pmc pass
# each (MI355X_MICROARCH.md, HBM: the two do not fit one pass; --kernel-trace only beside --pmc).   bash scripts/ubench/zlmo_stream_pmc.sh <tag> [dtype]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/zlmo_stream_pmc_${1:-a}
DT=${2:-f16}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python3 $ROOT/scripts/ubench/zlmo_stream.py --dtype $DT --reps 30"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o z -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o z -- $CMD > "$OUT/write.log" 2>&1
cd "$ROOT"
python3 - "$OUT" "$DT" <<'PY'
import csv, glob, json, re, sys, collections
out, dt = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for sub, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for f in glob.glob(out + f"/{sub}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name and "lc_" in r["Kernel_Name"][:80]:
                k = re.sub(r"^.*?(lc_[a-z0-9_]+).*$", r"\1", r["Kernel_Name"])
                acc[k][name].append(float(r["Counter_Value"]))
res = {}
for k, d in sorted(acc.items()):
    f = sum(d["FETCH_SIZE"]) / max(1, len(d["FETCH_SIZE"])); w = sum(d["WRITE_SIZE"]) / max(1, len(d["WRITE_SIZE"]))
    res[k] = {"launches": len(d["FETCH_SIZE"]), "fetch_kb_raw": round(f, 1), "write_kb_raw": round(w, 1)}
    print("%-40s n=%3d  FETCH_SIZE %10.1f KB   WRITE_SIZE %10.1f KB" % (k, len(d["FETCH_SIZE"]), f, w))
json.dump({"dtype": dt, "shape": "B=32 C=21 128x128 stride 3 (zlmo), cold inputs (ring of buffer sets > 256 MiB)", "kernels": res,
           "note": "raw counters, KB per launch (mean); gfx950: FETCH_SIZE reports half the bytes of 16 B/lane streaming reads, other widths uncalibrated (MI355X_MICROARCH.md, HBM)"},
          open(out + "/zlmo_stream_pmc.json", "w"), indent=1)
PY
