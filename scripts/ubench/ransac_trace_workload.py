"""Workload for a kernel trace of the split RANSAC form (rocprofv3 --kernel-trace ... -- python3 scripts/ubench/ransac_trace_workload.py;
summarise with scripts/ubench/trace_summary.py DIR ransac)."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from lc_amd import synth
from lc_amd.pnp import gpu_solver
dev = torch.device("cuda:0")
for (B, N) in ((64, 1024), (256, 1024), (256, 64)):
    bt = synth.make_batch(B, N, seed=2, outlier_frac=0.2)
    K, X, U = bt["K"].to(dev), bt["pts3d"].to(dev), bt["pts2d"].to(dev)
    for _ in range(20):
        gpu_solver.solve_device(K, X, U, reprojectionError=3.0, refine=False, split=True)
    torch.cuda.synchronize()
