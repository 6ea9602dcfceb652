#!/usr/bin/env python3
"""Per-kernel averages of the instruction-fetch counter passes written by scripts/profile_icache.sh (rocprofv3 counter_collection CSVs)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
WANT = ("lc_pnp_lm_chain_kernel", "lc_pose_unit_dense_kernel", "lc_pose_unit_kernel", "lc_pnp_lm_wide_kernel", "lc_cov_loss_tiled_kernel")
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))  # kernel -> counter -> [sum, launches]
for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if not any(w in k for w in WANT):
            continue
        k = k.replace("lc::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        key = f"{k} [grid {r.get('Grid_Size', '?')}]"
        a = acc[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
print("| kernel | counter | launches | mean per launch |\n|---|---|---|---|")
for k in sorted(acc):
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print(f"| {k} | {c} | {n} | {s / n:.4g} |")
print()
print("| kernel | ICACHE hit rate | misses per 1000 requests | SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES | SQ_IFETCH per wave |\n|---|---|---|---|---|")
for k in sorted(acc):
    m = {c: v[0] / v[1] for c, v in acc[k].items()}
    req, hit, mis = m.get("SQC_ICACHE_REQ"), m.get("SQC_ICACHE_HITS"), m.get("SQC_ICACHE_MISSES")
    wi, wc, ifetch, waves = m.get("SQ_WAIT_INST_ANY"), m.get("SQ_WAVE_CYCLES"), m.get("SQ_IFETCH"), m.get("SQ_WAVES")
    f = lambda a, b, s=1.0: f"{s * a / b:.4g}" if (a is not None and b) else "n/a"  # noqa: E731
    print(f"| {k} | {f(hit, req)} | {f(mis, req, 1000.0)} | {f(wi, wc)} | {f(ifetch, waves)} |")
