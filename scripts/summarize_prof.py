#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + per-dispatch PMC) into a small markdown table."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
git_sha = sys.argv[2] if len(sys.argv) > 2 else "unrecorded"
bench_cmd = sys.argv[3] if len(sys.argv) > 3 else None


def find(pattern):
    return sorted(glob.glob(os.path.join(out, pattern), recursive=True))


def short(name, grid=None):
    import re

    m = re.search(r"lc_pnp_lm_wide_kernel<[^>]*>", name)
    if m:
        return m.group(0).replace(" ", "")  # <REG, OPTS, points per thread in registers>: glmo's N = 1024 -> 4, zlmo's N = 1849 -> 8
    if "lc_cov_loss_tiled_kernel" in name:  # one name for every dense shape: told apart by the grid (workgroups of 256 threads)
        return "lc_cov_loss_tiled_kernel" + (f"[{int(grid) // 256} workgroups]" if grid else "")
    for key in ("lc_pose_unit_kernel", "lc_cov_loss_kernel", "lc_pnp_lm_kernel", "lc_head_fwd_wave64_kernel", "lc_head_fwd_rows_kernel", "lc_head_fwd_kernel", "lc_head_bwd_kernel",
                "lc_scale_rows_kernel", "lc_dense"):
        if key in name:
            return key
    return name[:60]


print("# rocprofv3 summary\n")

# what the code object tells the hardware to allocate per kernel (scripts/kernel_resources.py) -- the figure that bounds occupancy;
# rocprofv3's per-dispatch VGPR_Count column is NOT it (it read 128 for kernels that allocate 256)
try:
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from kernel_resources import kernel_resources

    RES = kernel_resources()
except Exception as e:  # noqa: BLE001
    RES = {}
    print(f"(kernel resources unavailable: {e})\n")
if RES:
    print("## kernel resources (from the code object in liblc_amd.so)\n")
    print("| kernel | arch VGPRs | AGPRs | SGPRs | LDS B/workgroup | scratch B/lane | waves/SIMD by registers |")
    print("|---|---|---|---|---|---|---|")
    for name, d in sorted(RES.items()):
        if any(k in name for k in ("pose_unit", "pnp_lm", "cov_loss")):
            print(f"| `{name}` | {d['arch_vgpr_count']} | {d.get('agpr_count', 0)} | {d.get('sgpr_count', 0)} | {d.get('group_segment_fixed_size', 0)} | "
                  f"{d.get('private_segment_fixed_size', 0)} | {d['waves_per_simd_by_registers']} |")
    print()


def unit_vgprs(B):
    """Allocation of the pose-unit instantiation a batch of B poses launches (2 B workgroups; <= 1024 -> the latency build)."""
    for name, d in RES.items():
        if "lc_pose_unit_kernelILi%dE" % (1 if 2 * B <= 1024 else 2) in name:
            return d.get("vgpr_count")
    return ""
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
            acc[short(r["Kernel_Name"], r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
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
                acc[short(r["Kernel_Name"], r.get("Grid_Size"))].append(float(r["Counter_Value"]))
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
        acc[short(r["Kernel_Name"], r.get("Grid_Size"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if k.startswith("lc_"):
            traffic.setdefault(k, {})["sq"] = {c: sum(v) / len(v) for c, v in d.items()}
traffic["_meta"] = {"git_sha": git_sha, "command": bench_cmd, "profiler": "rocprofv3 --kernel-trace --pmc <one counter group per pass>",
                    "round": os.path.basename(out.rstrip("/")).replace("prof_", "")}
json.dump(traffic, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)

# ---- batch-size sweep of the pose unit: dispatches grouped by grid size (grid = 2 B workgroups of 64 threads) ----
sweep = {}
for f in find("sweep_trace/**/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "lc_pose_unit_kernel" in r["Kernel_Name"]:
            B = int(r["Grid_Size_X"]) // 128
            sweep.setdefault(B, {"ns": [], "vgpr": r.get("VGPR_Count")})["ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
sq = {}
for f in find("sweep_pmc_sq/**/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "lc_pose_unit_kernel" in r["Kernel_Name"]:
            B = int(r["Grid_Size"]) // 128
            sq.setdefault(B, defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
ev = {}
p_ev = os.path.join(out, "sweep_events.jsonl")
if os.path.exists(p_ev):
    for ln in open(p_ev):
        if ln.startswith("{"):
            d = json.loads(ln)
            ev[d["B"]] = d
if sweep or ev:
    print("## pose-unit launch over batch size (N = 64): rocprofv3 kernel-trace duration per dispatch, SQ counters per dispatch\n")
    print("| B | VGPRs | avg us (trace) | min us | poses/s (trace avg) | us (events, un-profiled) | VALU insts / pose | VALU-issue SIMD-cycles / pose | VALU-issue bound poses/s | achieved / bound | wave-cycles VALU-active |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    table = {}
    for B in sorted(set(sweep) | set(ev)):
        ns = sorted(sweep.get(B, {}).get("ns", []))
        ns = ns[len(ns) // 5:] if len(ns) > 5 else ns  # drop nothing systematic; the first launches are not special under the trace
        avg = sum(ns) / len(ns) / 1e3 if ns else float("nan")
        c = {k: sum(v) / len(v) for k, v in sq.get(B, {}).items()}
        cyc = 4.0 * c["SQ_ACTIVE_INST_VALU"] / B if c.get("SQ_ACTIVE_INST_VALU") else float("nan")
        bound = 1024 * 2.4e9 / cyc if cyc == cyc else float("nan")
        pps = B / (avg * 1e-6) if avg == avg else float("nan")
        act = c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"] if c.get("SQ_WAVE_CYCLES") else float("nan")
        table[B] = {"avg_us": avg, "poses_per_s": pps, "valu_insts_per_pose": c.get("SQ_INSTS_VALU", float("nan")) / B, "simd_cycles_per_pose": cyc,
                    "valu_bound_poses_per_s": bound, "frac": pps / bound if bound == bound else None, "events_us": ev.get(B, {}).get("us_per_launch")}
        print(f"| {B} | {unit_vgprs(B)} | {avg:.1f} | {min(ns) / 1e3 if ns else float('nan'):.1f} | {pps:.3g} | {ev.get(B, {}).get('us_per_launch', float('nan'))} | "
              f"{c.get('SQ_INSTS_VALU', float('nan')) / B:.0f} | {cyc:.0f} | {bound:.3g} | {pps / bound if bound == bound else float('nan'):.3f} | {act:.2f} |")
    json.dump({"_meta": traffic["_meta"], "sweep": table}, open(os.path.join(out, "sweep_pose_unit.json"), "w"), indent=1)
    print()
