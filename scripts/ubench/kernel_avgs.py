#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --kernel-trace CSV (newest *_kernel_trace.csv under the directory given): name, launches, avg us."""
import csv
import glob
import os
import sys
from collections import defaultdict

files = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
if not files:
    sys.exit("no kernel trace under " + sys.argv[1])
tot, cnt = defaultdict(float), defaultdict(int)
for r in csv.DictReader(open(files[-1])):
    k = r["Kernel_Name"]
    tot[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    cnt[k] += 1
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for k in sorted(tot, key=tot.get, reverse=True):
    if cnt[k] > skip:
        print(f"{k[:110]:110s} {cnt[k]:6d} x {tot[k] / cnt[k]:8.2f} us")
