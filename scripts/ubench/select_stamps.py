"""Phase clock of lc_dense_frontend_select_kernel's wide path (16 384 candidates per object; workgroup 0, thread 0): diagnostic build
-DLC_SELECT_STAMPS, s_memtime ticks and each phase's share (the tick is not a documented unit here: quote shares, and kernel times from rocprofv3)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lc_amd import build  # noqa: E402

LIB = build.variant_path("selstamps")
if "--build" in sys.argv or not os.path.exists(LIB):
    build.build_variant("selstamps", ["-DLC_SELECT_STAMPS"])
    if "--build" in sys.argv:
        sys.exit(0)
os.environ["LC_AMD_LIB"] = LIB

import numpy as np  # noqa: E402
import torch  # noqa: E402

from lc_amd import _lib, synth  # noqa: E402
from lc_amd.dense import dense_front_end_select  # noqa: E402

SPLIT_NAMES = ["own entries and the log-sum-exp share requested, wavefront pairs formed", "the pairs and visible counts meet", "weights, keys",
               "radix select: four passes, the bins meet in each", "keep flags, local offsets; the counts meet", "rows written, padding"]
NAMES = ["48 requests + log-sum-exp over 32768 logits", "weights, visibility bits, keys to LDS", "quantile threshold (radix select)",
         "keep flags, 16 x 16 count table, scan (3 barriers)", "xyz of the entries, survivors written, padding"]
dev = torch.device("cuda:0")
lib = _lib.load()
fn = lib.lc_debug_select_stamps
fn.argtypes = [ctypes.c_void_p]
cfg, gt, out = synth.test_time_inputs("zlmo", B=64, seed=3)
xyz = torch.randn(64, 3, 128, 128).to(dev)
wl, ws, vl = out["xyz_weight_logits"].to(dev), out["xyz_weights_scale"].to(dev), out["msk_vis_logits"].to(dev)
if os.environ.get("MAPS") == "bf16":
    xyz, wl, vl = xyz.to(torch.bfloat16), wl.to(torch.bfloat16), vl.to(torch.bfloat16)
SPLIT = os.environ.get("LC_SELECT_SPLIT") == "1"  # four workgroups per object (part 0 of object 0 holds the clock)
rows = []
for it in range(24):
    for _ in range(2):
        dense_front_end_select(xyz, wl, ws, None, vl, "quantile_in_mask", quantile=0.2, sample=1, split=SPLIT)
    torch.cuda.synchronize()
    o = (ctypes.c_ulonglong * 10)()
    assert fn(o) == 0
    rows.append(np.diff(np.array(list(o)[:7 if SPLIT else 6], dtype=np.float64)))
d = np.median(np.array(rows), axis=0)
print("# scripts/ubench/select_stamps.py: lc_dense_frontend_select_kernel<.., true>, workgroup 0, median of 24 launches of 64 x 128x128, s_memtime ticks")
print("# (ONE wavefront's clock: its waits at the barriers are the other wavefronts' work -- scripts/ubench/select_skip_ab.py has the search's cost in kernel time)")
for n_, v in zip(SPLIT_NAMES if SPLIT else NAMES, d):
    print(f"  {n_:60s} {v:9.0f} ticks  {100 * v / d.sum():5.1f} %")
print(f"  {'total':60s} {d.sum():9.0f} ticks")
