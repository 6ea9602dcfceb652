#!/usr/bin/env python3
"""Per-kernel means of the SQ counter passes written by scripts/profile_chain_pmc.sh (rocprofv3 counter_collection CSVs), and what they say:
VALU-active share of the wave cycles, wave cycles waiting, wavefronts per launch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))  # kernel -> counter -> [sum, launches]
for f in sorted(glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "lc_" not in k:
            continue
        k = k.replace("lc::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_INSTS_LDS",
        "SQ_ACTIVE_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU", "GRBM_GUI_ACTIVE"]
print("Means per launch (rocprofv3 --pmc, one pass per group; counters summed over the chip as rocprofv3 reports them).\n")
print("| kernel | " + " | ".join(cols) + " |\n|---|" + "---|" * len(cols))
for k in sorted(acc):
    m = {c: v[0] / v[1] for c, v in acc[k].items()}
    print(f"| {k} | " + " | ".join(f"{m[c]:.4g}" if c in m else "n/a" for c in cols) + " |")
print("\n| kernel | wavefronts | VALU instructions per wavefront | VALU-active / wave cycles | waiting for an instruction / wave cycles | waiting for anything / wave cycles | "
      "LDS-active / wave cycles | wave cycles x 4 / GRBM_GUI_ACTIVE (wavefronts resident on average, in the counters' scope) |\n|---|---|---|---|---|---|---|---|")
for k in sorted(acc):
    m = {c: v[0] / v[1] for c, v in acc[k].items()}
    f = lambda a, b, s=1.0: f"{s * m[a] / m[b]:.3g}" if (a in m and b in m and m[b]) else "n/a"  # noqa: E731
    print(f"| {k} | {m.get('SQ_WAVES', float('nan')):.0f} | {f('SQ_INSTS_VALU', 'SQ_WAVES')} | {f('SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES')} | {f('SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES')} | "
          f"{f('SQ_WAIT_ANY', 'SQ_WAVE_CYCLES')} | {f('SQ_ACTIVE_INST_LDS', 'SQ_WAVE_CYCLES')} | {f('SQ_WAVE_CYCLES', 'GRBM_GUI_ACTIVE', 4.0)} |")
print("\n(SQ_WAVE_CYCLES, SQ_ACTIVE_INST_* and SQ_WAIT_* all count quad-cycles on gfx950 (MI355X_MICROARCH.md): their ratios are shares of the wave cycles as they stand; "
      "GRBM_GUI_ACTIVE counts cycles.)")
