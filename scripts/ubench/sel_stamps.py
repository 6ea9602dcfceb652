"""Phase clock of lc_ransac_select_kernel (workgroup 0, thread 0): diagnostic build -DLC_P3P_STAMPS, shader cycles (s_memtime)."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from lc_amd import build  # noqa: E402

LIB = build.variant_path("p3pstamps")
if "--build" in sys.argv or not os.path.exists(LIB):
    build.build_variant("p3pstamps", ["-DLC_P3P_STAMPS"])
    if "--build" in sys.argv:
        sys.exit(0)
os.environ["LC_AMD_LIB"] = LIB

import numpy as np  # noqa: E402
import torch  # noqa: E402

from lc_amd import _lib, synth  # noqa: E402
from lc_amd.pnp import gpu_solver  # noqa: E402

NAMES = ["count, points and partials requested .. column sums", "the parts' bests meet (barrier)", "inlier flags of the part's points (barrier)",
               "the parts' counts meet (barrier)", "rows written", "scalars, padding"]
NAMES = ["count, then all requests .. partial sums", "arg-max over lanes and waves (barrier)", "winner's pose through LDS (barrier)",
         "inlier mask + compaction + state outputs", "inlier count", "padding + selection count"]
dev = torch.device("cuda:0")
lib = _lib.load()
fn = lib.lc_debug_sel_stamps
fn.argtypes = [ctypes.c_void_p]
rows = []
B, N = 64, int(os.environ.get("LC_SEL_N", "1024"))
LOW, HIGH = (300, 560) if N == 1024 else (int(os.environ.get("LC_SEL_LOW", N // 6)), int(os.environ.get("LC_SEL_HIGH", N // 4)))
g = torch.Generator().manual_seed(0)
w = (torch.rand(B, N, 2, generator=g) + 0.1).to(dev)
for seed in range(24 if N == 1024 else 8):
    bt = {k: v.to(dev) for k, v in synth.make_batch(B, N, seed=seed, outlier_frac=0.3, noise_px=0.7).items()}
    counts = torch.randint(LOW, HIGH, (B,), generator=g).to(torch.int32).to(dev)
    for _ in range(2):
        gpu_solver.solve_device(bt["K"], bt["pts3d"], bt["pts2d"], counts, reprojectionError=3.0, refine=False, split=True, select=dict(weights=w))
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    assert fn(out) == 0
    rows.append(np.diff(np.array(list(out)[:7], dtype=np.float64)))
d = np.median(np.array(rows), axis=0)
print(f"# scripts/ubench/sel_stamps.py: lc_ransac_select_kernel, workgroup 0, median over 24 batches of {B} x {N} ({LOW}-{HIGH} used), s_memtime counts")
for n_, v in zip(NAMES, d):
    print(f"  {n_:40s} {v:9.0f}  {100 * v / d.sum():5.1f} %")
print(f"  {'total':40s} {d.sum():9.0f}")
