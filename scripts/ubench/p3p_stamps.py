"""Phase clock of the RANSAC hypothesis kernel (one wavefront = 64 P3P hypotheses): diagnostic build -DLC_P3P_STAMPS, shader cycles.

    python scripts/ubench/p3p_stamps.py            # builds build/variants/liblc_amd_p3pstamps.so here, run on the GPU box
"""
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

NAMES = ["launch .. sample indices", "point loads, bearings, X^-1, determinants", "cubic roots", "root loop (plane pairs, candidates, 4th-point choice)",
         "Gauss-Newton polish of the chosen candidate + pose", "store"]


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    fn = lib.lc_debug_p3p_stamps
    fn.argtypes = [ctypes.c_void_p]
    rows = []
    for seed in range(24):
        bt = synth.make_batch(1, 1024, seed=seed, outlier_frac=0.2)
        K, X, U = bt["K"].to(dev), bt["pts3d"].to(dev), bt["pts2d"].to(dev)
        for _ in range(2):
            gpu_solver.solve_device(K, X, U, reprojectionError=3.0, refine=False, iterations=64, seed=seed)
        torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 7)()
        assert fn(out) == 0
        st = np.array(list(out), dtype=np.float64)
        rows.append(np.diff(st))
    d = np.median(np.array(rows), axis=0)
    print("# scripts/ubench/p3p_stamps.py: one wavefront of 64 hypotheses, median over 24 poses, s_memtime counts (shader cycles: 29.2k of them are the 13.3 us of the PnP kernel)")
    for n, v in zip(NAMES, d):
        print(f"  {n:48s} {v:9.0f}  {100 * v / d.sum():5.1f} %")
    print(f"  {'total':48s} {d.sum():9.0f}")


if __name__ == "__main__":
    main()
