#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + per-dispatch PMC) into a small markdown table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def short(name):
    for key in ("lc_pose_unit_kernel", "lc_cov_loss_kernel", "lc_pnp_lm_kernel", "lc_head_fwd_wave64_kernel", "lc_head_fwd_rows_kernel", "lc_head_fwd_kernel", "lc_head_bwd_kernel",
                "lc_scale_rows_kernel", "lc_dense"):
        if key in name:
            return key
    return name[:60]


print("# rocprofv3 summary\n")
for sub, title in (("trace", "bench.py (pose unit, B=256 N=64)"), ("head_trace", "bench_head.py (keypoint head, 256x64x64x64)")):
    for f in find(f"{sub}/**/*kernel_stats.csv"):
        print(f"## kernel stats: {title}\n")
        print("| kernel | calls | total ns | avg ns | min ns | max ns | % |")
        print("|---|---|---|---|---|---|---|")
        for r in csv.DictReader(open(f)):
            print(f"| {short(r['Name'])} | {r['Calls']} | {r['TotalDurationNs']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {float(r['Percentage']):.2f} |")
        print()

for sub, title in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_sq", "SQ counters"), ("head_pmc_fetch", "head FETCH_SIZE"),
                   ("head_pmc_write", "head WRITE_SIZE")):
    for f in find(f"{sub}/**/*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        print(f"## {title} (mean per dispatch)\n")
        print("| kernel | counter | dispatches | mean | ")
        print("|---|---|---|---|")
        for k, d in acc.items():
            for c, v in d.items():
                print(f"| {k} | {c} | {len(v)} | {sum(v) / len(v):.4g} |")
        print()


# machine-readable HBM traffic per launch for bench.py's roofline.traffic (bytes; FETCH_SIZE/WRITE_SIZE are in KB)
import json
traffic = {}
for sub, key in (("pmc_fetch", "fetch"), ("pmc_write", "write"), ("head_pmc_fetch", "fetch"), ("head_pmc_write", "write")):
    for f in find(f"{sub}/**/*counter_collection.csv"):
        acc = defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            if k.startswith("lc_"):
                traffic.setdefault(k, {})[key + "_kb_raw"] = sum(v) / len(v)
for k, d in traffic.items():
    wide = k.startswith("lc_head")  # 16 B/lane streaming reads: gfx950 FETCH_SIZE counts half the bytes (MI355X_MICROARCH.md, HBM)
    d["fetch_correction"] = 2.0 if wide else 1.0
    d["bytes_per_launch"] = (d.get("fetch_kb_raw", 0.0) * d["fetch_correction"] + d.get("write_kb_raw", 0.0)) * 1024
    d["note"] = ("FETCH_SIZE x2 (gfx950 correction for 16 B/lane streams)" if wide else
                 "raw counters; 8-12 B/lane accesses are uncalibrated on gfx950 (MI355X_MICROARCH.md, HBM)")
# issue counters of the latency/VALU-bound kernels (mean per dispatch), for bench.py's roofline.valu
for f in find("pmc_sq/**/*counter_collection.csv"):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if k.startswith("lc_"):
            traffic.setdefault(k, {})["sq"] = {c: sum(v) / len(v) for c, v in d.items()}
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
