#!/bin/bash
# rocprofv3 kernel-trace averages of a python script's lc_* kernels:  bash scripts/ubench/kernel_times.sh <script.py> [filter regex]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -o kt -- python3 "$ROOT/$1" > /tmp/kt.log 2>&1
python3 - "${2:-.}" <<'PY'
import csv, glob, re, sys
f = glob.glob("/tmp/kt/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if ("lc_" in n) and re.search(sys.argv[1], n):
        n = n.replace("lc::(anonymous namespace)::", "").replace("_ZN2lc12_GLOBAL__N_1", "")
        print(f"{n[:100]:100s} {r['Calls']:>6s} x {float(r['AverageNs']) / 1e3:9.2f} us")
PY
