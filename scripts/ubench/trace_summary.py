"""Median duration per (kernel, grid) from a rocprofv3 --kernel-trace --output-format csv directory:  python trace_summary.py DIR [substr]"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if sub in n:
        m = re.search(r"(lc_\w+)", n)
        d[(m.group(1) if m else n[:44], r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Workgroup_Size_X") or r.get("Workgroup_Size"))].append(
            int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    v.sort()
    print("%-46s grid %-8s wg %-5s calls %4d  median %8.2f us  min %8.2f" % (k[0], k[1], k[2], len(v), v[len(v) // 2] / 1e3, v[0] / 1e3))
